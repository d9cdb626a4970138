"""Pins the oracle (CPU restatement, oracle/restate) bit-for-bit to golden vectors produced by the
reference's own unmodified GLSL running on Mesa llvmpipe (tests/golden/make_golden.py).

CPU-only: no GPU, no reference checkout needed."""
import glob
import os

import numpy as np
import pytest

from gpuart_amd import synth_scenes as S
from oracle import oracle as O
from tests.util import fuzz_case_setup, row_checksums, GOLDEN, assert_bits, frame_golden_params, golden, pad4, scene


def test_hash_random():
    g = golden("hash")
    assert_bits(O.random(g["x"]), g["out"], "random(float/vec2/vec3/vec4)")


def test_llvmpipe_sin_cos_pow():
    g = golden("llvmpipe_math")
    x = np.zeros((len(g["x"]), 4), np.float32)
    x[:, 0] = g["x"][:, 0]
    sc = O.sincos(x)
    assert_bits(sc[:, 0], g["out"][:, 0], "sin")
    assert_bits(sc[:, 1], g["out"][:, 1], "cos")
    x[:, 0] = g["x"][:, 1]
    assert_bits(O.pow16(x)[:, 0], g["out"][:, 2], "pow(x,16)")
    assert_bits(np.sqrt(g["x"][:, 1]), g["out"][:, 3], "sqrt is IEEE")


def test_hemisphere_sampler():
    g = golden("hemisphere")
    assert_bits(O.hemisphere(pad4(g["v"]), pad4(g["ri"]))[:, :3], g["out"], "GetRandomHemisphereDirection")


def test_inside_cone_sampler():
    g = golden("inside_cone")
    ha = np.float32(10) * np.float32(3.14159) / np.float32(180)
    assert_bits(O.inside_cone(pad4(g["v"]), pad4(g["normal"]), pad4(g["ri"]), ha)[:, :3], g["out"],
                "GetRandomDirectionInsideCone")


def _hit(o):
    return np.concatenate(o, 1)


def test_sphere():
    g = golden("sphere")
    assert (g["o0"][:, 0] > 0).mean() > 0.2
    assert_bits(_hit(O.sphere(pad4(g["rs"]), pad4(g["rd"]), g["sph"])), _hit([g["o0"], g["o1"]]), "SphereIntersection")


def test_disc():
    g = golden("disc")
    assert (g["o0"][:, 0] > 0).mean() > 0.2
    assert_bits(_hit(O.disc(pad4(g["rs"]), pad4(g["rd"]), g["cr"], pad4(g["dn"]))), _hit([g["o0"], g["o1"]]),
                "DiscIntersection")


def test_triangle():
    g = golden("triangle")
    assert (g["o0"][:, 0] > 0).mean() > 0.2
    got = O.triangle(*[pad4(g[k]) for k in ["rs", "rd", "v0", "v1", "v2"]])
    assert_bits(_hit(got), _hit([g["o0"], g["o1"]]), "TriangleIntersection")


def test_cone():
    g = golden("cone")
    q = g["quads"]
    assert (g["o0"][:, 0] > 0).mean() > 0.2
    o0, o1 = O.cone(pad4(g["rs"]), pad4(g["rd"]), q[:, 0:4], q[:, 4:8], q[:, 8:12], q[:, 12:16])
    m = o0[:, 0] < 1e-4  # CheckBVHPrimitiveIntersection's visibility cut, applied in the probe too
    o0[m] = [-1, 0, 0, 0]
    o1[m] = 0
    assert_bits(_hit([o0, o1]), _hit([g["o0"], g["o1"]]), "ConeIntersection")


def _cut(o0, o1):
    """CheckBVHPrimitiveIntersection's visibility cut + the probe's output shape (pos > 0: pos, P, N; else pos alone)."""
    o0, o1 = o0.copy(), o1.copy()
    o0[o0[:, 0] < 1e-4, 0] = -1
    keep = o0[:, 0] > 0
    o0[~keep, 1:] = 0
    o1[~keep] = 0
    return np.concatenate([o0, o1], 1)


def _bits_or_nan(got, exp, what):
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    bad = ~same.all(1)
    assert not bad.any(), "%s: %d of %d rows differ; first: row %d got %s expected %s" % (what, int(bad.sum()), len(bad), int(np.nonzero(bad)[0][0]),
                                                                                         got[bad][0], exp[bad][0])


@pytest.mark.parametrize("name", ["scene_pc", "wild_42874", "wild_7", "wild2_5", "pc_min5", "pc_levels4", "p_root_leaf", "soup_levels6"])
def test_traversal_with_hostile_rays_and_trees(name):
    """CheckIntersectionInclUserSphere with hostile rays on a regular tree and with regular + hostile rays on the trees of wild
    scenes (irregular boxes, a 200-level chain) and on trees with other leaf sizes than the default build's (5 primitives per
    leaf, depth limits 4 / 6 / 1 = everything in the root leaf), user sphere of radius 0.25, against the reference's GLSL on llvmpipe."""
    g = golden("traverse_wild_" + name)
    o0, o1 = O.traverse(g["tree"], pad4(g["rs"]), pad4(g["rd"]), (-0.4, 0.0, 0.2, 0.25))
    got, exp = np.concatenate([o0, o1], 1), np.concatenate([g["o0"], g["o1"]], 1)
    same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
    bad = ~same.all(1)
    assert not bad.any(), "%d of %d rays differ; first: ray %d o %s d %s got %s expected %s" % (
        int(bad.sum()), len(bad), int(np.nonzero(bad)[0][0]), g["rs"][bad][0], g["rd"][bad][0], got[bad][0], exp[bad][0])


def test_shading_functions_on_hostile_numbers():
    """random(), GetRandomHemisphereDirection, GetRandomDirectionInsideCone and GetSkyColor on NaN / infinite / huge / denormal
    inputs (what a path carries after bouncing off a wild primitive), against the reference's GLSL on llvmpipe."""
    def nanbits(got, exp, what):
        got, exp = np.asarray(got, np.float32), np.asarray(exp, np.float32)
        same = (got.view(np.uint32) == exp.view(np.uint32)) | (np.isnan(got) & np.isnan(exp))
        bad = ~same.reshape(len(got), -1).all(1)
        assert not bad.any(), "%s: %d of %d rows differ; first: row %d got %s expected %s" % (what, int(bad.sum()), len(bad),
                                                                                             int(np.nonzero(bad)[0][0]), got[bad][0], exp[bad][0])
    g = golden("hash_wild")
    nanbits(O.random(g["x"]), g["out"], "random")
    g = golden("hemisphere_wild")
    nanbits(O.hemisphere(pad4(g["v"]), pad4(g["ri"]))[:, :3], g["out"], "GetRandomHemisphereDirection")
    g = golden("inside_cone_wild")
    ha = np.float32(10) * np.float32(3.14159) / np.float32(180)  # the probe's constant, folded in float as GLSL folds it
    nanbits(O.inside_cone(pad4(g["v"]), pad4(g["normal"]), pad4(g["ri"]), ha)[:, :3], g["out"], "GetRandomDirectionInsideCone")
    g = golden("sky_wild")
    nanbits(O.sky(pad4(g["dir"]), g["sun_dir_alt"])[:, :3], g["out"], "GetSkyColor")


def test_intersectors_on_hostile_numbers():
    """The four intersectors with NaN / +-inf / +-1e30 / denormals / negative radii / the reference's magic numbers in the
    primitive and in the ray, against the reference's GLSL on llvmpipe (tests/golden/make_golden.py intersect_wild)."""
    g = golden("sphere_wild")
    assert 0.05 < (g["o0"][:, 0] > 0).mean()
    _bits_or_nan(_cut(*O.sphere(pad4(g["rs"]), pad4(g["rd"]), g["sph"])), np.concatenate([g["o0"], g["o1"]], 1), "SphereIntersection")
    g = golden("disc_wild")
    _bits_or_nan(_cut(*O.disc(pad4(g["rs"]), pad4(g["rd"]), g["cr"], pad4(g["dn"]))), np.concatenate([g["o0"], g["o1"]], 1), "DiscIntersection")
    g = golden("triangle_wild")
    _bits_or_nan(_cut(*O.triangle(*[pad4(g[k]) for k in ["rs", "rd", "v0", "v1", "v2"]])), np.concatenate([g["o0"], g["o1"]], 1), "TriangleIntersection")
    g = golden("cone_wild")
    q = g["quads"]
    _bits_or_nan(_cut(*O.cone(pad4(g["rs"]), pad4(g["rd"]), q[:, 0:4], q[:, 4:8], q[:, 8:12], q[:, 12:16])), np.concatenate([g["o0"], g["o1"]], 1),
                 "ConeIntersection")


def test_aabb():
    g = golden("aabb")
    assert 0.2 < g["out"][:, 0].mean() < 0.9
    assert_bits(O.aabb(pad4(g["rs"]), pad4(g["rd"]), pad4(g["bmin"]), pad4(g["bmax"]))[:, :2], g["out"], "IntersectsAABB")


def test_aabb_irregular_boxes():
    """Boxes with an inverted, NaN or infinite axis and rays with zero / infinite / NaN components, against the reference's
    IntersectsAABB on llvmpipe (tests/golden/make_golden.py aabb_irregular): a box with ONE irregular axis can be hit through
    that axis's planes, a plane at +-inf is an intersection that leaves the entry at 1e19."""
    g = golden("aabb_irregular")
    assert_bits(O.aabb(pad4(g["rs"]), pad4(g["rd"]), pad4(g["bmin"]), pad4(g["bmax"]))[:, :2], g["out"], "IntersectsAABB, irregular boxes")
    hit = g["out"][:, 0] > 0
    lo, hi = g["bmin"], g["bmax"]
    one_bad = (~(lo <= hi)).sum(1) == 1
    assert (hit & one_bad).sum() > 50, "the fixture must contain hits through the planes of a single irregular axis"



@pytest.mark.parametrize("k", [0, 1, 2])
def test_sky(k):
    g = golden("sky_%d" % k)
    assert_bits(O.sky(pad4(g["dir"]), g["sun_dir_alt"])[:, :3], g["out"], "GetSkyColor")


def test_quad_uv_interpolation():
    """vertex.glsl UV as llvmpipe rasterises the full-screen quad, 1x1 up to 7680x4320."""
    g = golden("uv")
    cam = np.zeros(13, np.float32)
    cam[3:6] = 0
    cam[6:9] = [1, 0, 0]   # rstart = (u, v, 0): BottomLeft 0, DeltaHorz x, DeltaVert y
    cam[9:12] = [0, 1, 0]
    cam[0:3] = [0.5, 0.5, -1]
    for key in g.files:
        kind, dims = key.split("_")
        W, H = map(int, dims.split("x"))
        if kind == "full":
            rs, _ = O.cam_rays(cam, W, H)
            assert_bits(rs[..., :2].reshape(-1, 2), g[key].reshape(-1, 2), "UV " + dims)
        elif kind == "uv":
            xy = g["xy_" + dims]
            got = O.pixel_uv(xy, W, H)
            assert_bits(got, g[key], "UV samples " + dims)


@pytest.mark.parametrize("name", ["camrays_64x36", "camrays_37x23"])
def test_camera_rays(name):
    g = golden(name)
    H, W = g["rstart"].shape[:2]
    rs, rd = O.cam_rays(g["cam"], W, H)
    assert_bits(rs[..., :3].reshape(-1, 3), g["rstart"].reshape(-1, 3), "cam_init rstart")
    assert_bits(rd[..., :3].reshape(-1, 3), g["rdir"].reshape(-1, 3), "cam_init rdir")


@pytest.mark.parametrize("name", ["box", "scene_pc", "scene_d"])
def test_traversal(name):
    from gpuart_amd import synth_scenes as S
    g = golden("traverse_" + name)
    tree, _ = O.build_bvh(scene(name))
    assert 0.2 < (g["o1"][:, 3] >= 0).mean() < 0.95
    o0, o1 = O.traverse(tree, pad4(g["rs"]), pad4(g["rd"]), S.USER_SPHERE)
    assert_bits(_hit([o0, o1]), _hit([g["o0"], g["o1"]]), "closest hit, primary rays")
    o0, o1 = O.traverse(tree, pad4(g["rs2"]), pad4(g["rd2"]), S.USER_SPHERE)
    assert_bits(_hit([o0, o1]), _hit([g["s0"], g["s1"]]), "closest hit, secondary + sun rays")


def test_rays_through_boxes_that_report_their_exit_as_their_entry():
    """Four closest-hit queries (tools/order_rays.py found them on the GPU among 1.7e9) on which a box of the tree is "hit" only
    through the face the ray LEAVES by — the face it enters by fails its own bounds test by rounding at an edge —, so that its
    entry parameter lies beyond the primitives inside and the answer depends on the reference's visiting order: the oracle's
    walk must give what the reference's GLSL gives on llvmpipe (make_golden.py order_rays)."""
    from gpuart_amd import synth_scenes as S
    g = golden("order_rays")
    for name, descs in (("cfg3", S.scene_d()), ("tree", S.tree_scene())):
        tree, _ = O.build_bvh(descs)
        o0, o1 = O.traverse(tree, g[name + "_rs"], g[name + "_rd"], (0, 0, 0, 0))
        assert_bits(np.concatenate([o0, o1], 1), np.concatenate([g[name + "_o0"], g[name + "_o1"]], 1), "order rays " + name)


def test_phantom_hits_of_grazing_triangles():
    """tests/golden/order_adversary.npz (make_golden.py order_adversary): 16 rays that graze a triangle by ~1e-6 rad; the reference's
    GLSL returns a phantom hit of that triangle — a small dyadic parameter from two cancelled sums (shaders/triangle.glsl:50-76), far in
    front of the triangle's own box and in front of the disc that stands before it. The oracle's walk (the reference's order, the
    reference's arithmetic) must return the same bits; its intersector alone must show the same phantom. This fixture is why the
    product walks every tree in the reference's order (tests/test_gpu_parity.py::test_phantom_hits_are_why_nearest_first_is_opt_in)."""
    g = golden("order_adversary")
    n = int(g["n"])
    assert n >= 16
    for i in range(n):
        rs, rd = pad4(g["rs%d" % i][None]), pad4(g["rd%d" % i][None])
        o0, o1 = O.traverse(g["tree%d" % i], rs, rd, (0, 0, 0, 0))
        assert_bits(np.concatenate([o0, o1], 1), np.concatenate([g["o0_%d" % i], g["o1_%d" % i]])[None], "adversary %d" % i)
        assert o1[0, 3] == 2.0 and o0[0, 0] < g["nf%d" % i]  # the triangle wins, in front of what a nearest-first walk finds


@pytest.mark.parametrize("k", [0, 1, 2])
def test_phantom_hits_in_whole_frames(k):
    """tests/golden/order_adversary_frame_k.npz (make_golden.py order_adversary_frames): the phantom hit end to end. The first-segment ray of
    one pixel of a 24x16 frame grazes a triangle laid under it; the reference's own path-tracing program shows the triangle's phantom in that
    pixel although a disc stands in front of the triangle. The oracle's passes must be the reference's frames bit for bit, and its first
    query of that pixel must return the phantom."""
    g = golden("order_adversary_frame_%d" % k)
    W, H = int(g["W"]), int(g["H"])
    cam, tree, seeds = g["cam"], g["tree"], g["seeds"]
    sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
    P = O.make_params(sun, S.SUN_ALTITUDE, True, S.USER_SPHERE, 0.0, 0, float(cam[12]), cam[0:3], int(g["max_segments"]), 0.01)
    acc = np.zeros((H, W, 4), np.float32)
    O.pt_pass(tree, cam, W, H, P, seeds[0], 1, acc)
    assert_bits(acc[..., :3].reshape(-1, 3), g["pt_pass1"].reshape(-1, 3), "first pass")
    O.pt_pass(tree, cam, W, H, P, seeds[1], 1, acc)
    assert_bits(acc[..., :3].reshape(-1, 3), g["pt_acc"].reshape(-1, 3), "two passes accumulated")
    px, py = (int(v) for v in g["pixel"])
    rs, rd = O.first_segment_rays(cam, W, H, P, seeds[0], 0)
    o0, o1 = O.traverse(tree, rs[py, px][None], rd[py, px][None], (0, 0, 0, 0))
    assert o1[0, 3] == 2.0 and o0[0, 0] == g["reference_t"] and o0[0, 0] < g["nearest_first_t"]


FRAMES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "frames_*.npz")))


@pytest.mark.parametrize("name", FRAMES)
def test_frames(name):
    """Whole frames: direct lighting, first PT pass, accumulated passes, 3 paths/pass — bit-exact."""
    g = golden(name)
    W, H = int(g["W"]), int(g["H"])
    tree, _ = O.build_bvh(scene(str(g["scene"])))
    cam = g["cam"]
    mk = frame_golden_params(O, g)
    nt = min(8, os.cpu_count() or 1)
    if "direct" in g:
        assert_bits(O.render_direct(tree, cam, W, H, mk(), nthreads=nt)[0][..., :3].reshape(-1, 3),
                    g["direct"].reshape(-1, 3), "direct lighting")
    if "direct_nosun" in g:
        assert_bits(O.render_direct(tree, cam, W, H, mk(False), nthreads=nt)[0][..., :3].reshape(-1, 3),
                    g["direct_nosun"].reshape(-1, 3), "direct lighting, sun off")
    seeds = g["seeds"]
    npass = int(g["npasses"]) if "npasses" in g else 2
    acc = np.zeros((H, W, 4), np.float32)
    for k in range(npass):
        O.pt_pass(tree, cam, W, H, mk(), seeds[k], 1, acc, nthreads=nt)
        if k == 0:
            assert_bits(acc[..., :3].reshape(-1, 3), g["pt_pass1"].reshape(-1, 3), "PT pass 1")
    assert_bits(acc[..., :3].reshape(-1, 3), g["pt_acc"].reshape(-1, 3), "PT accumulated")
    if "pt_3paths" in g:
        acc = np.zeros((H, W, 4), np.float32)
        O.pt_pass(tree, cam, W, H, mk(), seeds[0], 3, acc, nthreads=nt)
        assert_bits(acc[..., :3].reshape(-1, 3), g["pt_3paths"].reshape(-1, 3), "PT 3 paths in one pass")


def test_random_cases_vs_reference():
    """24 random cases (degenerate primitives, exact duplicates, random cameras, user-sphere modes, Sun on/off, depths
    1-8, 1-2 paths per pass) rendered by the reference's own shaders on llvmpipe (tests/golden/make_golden.py `fuzz`):
    the oracle reproduces direct lighting and three accumulated path-tracing passes bit for bit."""
    g = golden("fuzz_frames")
    for seed in g["cases"]:
        seed = int(seed)
        case, tree, cam, P, seeds = fuzz_case_setup(O, seed)
        W, H = case["W"], case["H"]
        assert_bits(O.render_direct(tree, cam, W, H, P)[0][..., :3].reshape(-1, 3), g["direct_%d" % seed].reshape(-1, 3),
                    "case %d direct lighting" % seed)
        acc = np.zeros((H, W, 4), np.float32)
        for k in range(case["passes"]):
            O.pt_pass(tree, cam, W, H, P, seeds[k], case["npaths"], acc)
        assert_bits(acc[..., :3].reshape(-1, 3), g["pt_acc_%d" % seed].reshape(-1, 3), "case %d path tracing" % seed)


@pytest.mark.parametrize("sc,tag", [("scene_d", "1080p"), ("scene_d", "4k"), ("scene_p", "1080p")])
def test_full_size_frame_vs_reference_checksums(sc, tag):
    """BASELINE cfg3 at FULL size (Scene D, 1920x1080, depth 8, benchmark camera) as rendered by the reference's shaders
    on llvmpipe, held as per-row checksums of the float bit patterns: the oracle's direct-lighting frame and its
    accumulator after one and two path-tracing passes give the same 3 x 1080 x 3 checksums. ("scene_p", "1080p") is BASELINE
    cfg2: Scene P (256 spheres + 16 discs), depth 4, default camera."""
    g = golden("fullsize_%s_%s" % (sc, tag))
    W, H = int(g["W"]), int(g["H"])
    tree, _ = O.build_bvh(scene(sc))
    cam = g["cam"]
    sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
    P = O.make_params(sun, S.SUN_ALTITUDE, True, S.USER_SPHERE, 0.0, 0, float(cam[12]), cam[0:3], int(g["max_segments"]), 0.01)
    nt = min(8, os.cpu_count() or 1)
    np.testing.assert_array_equal(row_checksums(O.render_direct(tree, cam, W, H, P, nthreads=nt)[0]), g["direct"])
    acc = np.zeros((H, W, 4), np.float32)
    for k in range(int(g["npasses"])):
        O.pt_pass(tree, cam, W, H, P, g["seeds"][k], 1, acc, nthreads=nt)
        np.testing.assert_array_equal(row_checksums(acc), g["pt_acc%d" % (k + 1)])


def test_randseed_sequence():
    """RandSeed quadruples of a default-seeded std::mt19937 (src/renderer.cpp:585-589); the first
    three were recorded from the reference's own binary run in SURVEY.md §8(a) row a18."""
    s = O.randseeds(3).view(np.uint32)
    assert [hex(v) for v in s[0]] == ["0x3f5091bb", "0x3e0aba7c", "0x3f67e1fb", "0x3f55c31f"]
    assert [hex(v) for v in s[1]] == ["0x3e0208d5", "0x3f7807b8", "0x3f69d300", "0x3e6256c0"]
    assert [hex(v) for v in s[2]] == ["0x3f21e24c", "0x3e9dc812", "0x3dc7c343", "0x3f0c16a6"]


def test_box_bvh_layout():
    """Compiled Box tree: canonical layout invariants (src/bvh.cpp:161-222)."""
    tree, depth = O.build_bvh(scene("box"))
    u = tree.view(np.uint32).astype(np.int64)
    assert tree.shape[0] * 16 == 1168  # "box_bvh.bin (1.1 KiB)" measured on the reference, SURVEY.md §8(c)
    assert u[2, 0] & (1 << 29) and not (u[2, 0] & (1 << 31))  # root: IS_ROOT, interior
    assert u[2, 1] == 3 and u[2, 3] == 0                        # lo child follows its parent; root parent = 0
    # every child points back to its parent; leaf prim counts add up to 9
    nprims, stack = 0, [0]
    while stack:
        a = stack.pop()
        fl = u[a + 2, 0]
        if fl & (1 << 31):
            nprims += fl & ~(7 << 29)
        else:
            lo, hi = int(u[a + 2, 1]), int(u[a + 2, 2])
            assert u[lo + 2, 3] == a and u[hi + 2, 3] == a
            assert u[lo + 2, 0] & (1 << 30) and not (u[hi + 2, 0] & (1 << 30))
            stack += [lo, hi]
    assert nprims == 9
