"""CPU-only tests of the product's host side: the C++ Renderer/Scene library (libgpuart.so) against the
oracle's host restatement, the loaders, and the C-ABI surface of libgpuart_hip.so (load + exports only —
no compute calls without a GPU)."""
import os
import re

import numpy as np
import pytest

from gpuart_amd import binding as B
from gpuart_amd import synth_scenes as S
from oracle import oracle as O
from tests.util import assert_bits, scene

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared(header):
    return sorted(set(re.findall(r"\b(gpuart_hip_[a-z_0-9]+)\s*\(", open(os.path.join(ROOT, "include", header)).read())))


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def test_c_abi_exports_every_declared_symbol():
    """The PRODUCT library (gpuart_amd/lib/libgpuart_hip.so: what libgpuart.so, gpuart_cli, bench.py and smoke() load) exports exactly what
    include/gpuart_hip.h declares — no test hook among it, and nothing else: a host program that defines `launch_run` or `check_shares` of
    its own must not be interposed (the link uses csrc/hip/exports.map). The library the test suite runs on (gpuart_amd/lib_test/, the same
    sources with -DGPUART_HIP_TEST_HOOKS) exports that plus exactly the hooks of include/gpuart_hip_test.h."""
    import ctypes as C
    product = os.path.join(ROOT, "gpuart_amd", "lib", "libgpuart_hip.so")
    names, hooks = _declared("gpuart_hip.h"), [n for n in _declared("gpuart_hip_test.h") if n.startswith("gpuart_hip_test_")]
    assert len(names) >= 25 and len(hooks) >= 15 and not [n for n in names if "_test_" in n]
    assert _exported(product) == names, (sorted(set(_exported(product)) - set(names)), sorted(set(names) - set(_exported(product))))
    L = C.CDLL(product)
    for n in names:
        assert hasattr(L, n), "libgpuart_hip.so does not export " + n
    tested = B.hip_lib()._name
    if os.path.dirname(tested) == os.path.join(ROOT, "gpuart_amd", "lib_test"):   # (an A/B run names its own directory: GPUART_LIBDIR)
        assert _exported(tested) == sorted(names + hooks)


def test_the_product_host_library_and_cli_are_bound_to_the_product_library():
    """gpuart_amd/lib/libgpuart.so and bin/gpuart_cli name libgpuart_hip.so as a dependency and find it beside themselves ($ORIGIN): never
    the test build."""
    import subprocess
    for path, rpath in ((os.path.join(ROOT, "gpuart_amd", "lib", "libgpuart.so"), "$ORIGIN"), (os.path.join(ROOT, "gpuart_amd", "bin", "gpuart_cli"), "$ORIGIN/../lib")):
        dyn = subprocess.run(["readelf", "-d", path], capture_output=True, text=True, check=True).stdout
        assert "libgpuart_hip.so" in dyn and rpath in dyn, dyn


def test_host_c_api_exports():
    L = B.host_lib()
    hdr = open(os.path.join(ROOT, "gpuart_amd", "csrc", "host", "capi.h")).read()
    for n in sorted(set(re.findall(r"\b(gpuart_[a-z_0-9]+)\s*\(", hdr))):
        assert hasattr(L, n), n


def test_product_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(B.HipError):
        B.Backend(0)
    cam = dict(S.DEFAULT_CAMERA); cam["dir"] = S.camera_dir(cam)
    with pytest.raises(B.HipError):
        B.Renderer(64, 64, cam)


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (no CPU fallback)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "gpuart_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")) or f == "Makefile":
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle|#include\s+[\"<].*oracle|liboracle", src, re.M), f


@pytest.mark.parametrize("name", ["box", "scene_pc", "scene_p", "scene_d"])
def test_bvh_bytes_equal_oracle(name):
    prims = scene(name)
    a, da = B.compile_bvh(prims)
    b, db = O.build_bvh(prims)
    assert da == db and a.shape == b.shape
    assert (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("levels,minprims", [(1024, 1), (1024, 4), (3, 2), (1, 2)])
def test_bvh_build_parameters(levels, minprims):
    prims = scene("scene_pc")
    a, _ = B.compile_bvh(prims, levels, minprims)
    b, _ = O.build_bvh(prims, levels, minprims)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_bvh_ties_and_duplicates():
    """Equal box centres exercise std::sort's unstable order; both sides must agree byte for byte."""
    rng = np.random.RandomState(9)
    prims = []
    for i in range(300):
        c = np.round(rng.uniform(-1, 1, 3) * 4) / 4  # coarse grid -> many identical centres
        prims.append((S.SPHERE, [float(c[0]), float(c[1]), float(c[2]), 0.1]))
    prims += prims[:50]
    a, _ = B.compile_bvh(prims)
    b, _ = O.build_bvh(prims)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("flavour", ["plain", "wild", "wild2"])
def test_bvh_bytes_equal_oracle_on_random_and_hostile_scenes(flavour):
    """The product's host build (Subdivide / Compile / StoreIntoBVH) against the oracle's on the cases of the parity soaks,
    hostile numbers included (NaN / +-inf / negative radii / magic numbers in the coordinates: NaN keys in the sort, inverted
    and infinite boxes, NaN payload bits): byte for byte. (4 500 cases of the same loop found no difference.)"""
    fn = {"plain": S.random_case, "wild": S.random_wild_case, "wild2": S.random_wild2_case}[flavour]
    for seed in range(300):
        prims = fn(seed)["prims"]
        a, da = B.compile_bvh(prims)
        b, db = O.build_bvh(prims)
        assert da == db and a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all(), "%s case %d" % (flavour, seed)


@pytest.mark.parametrize("fork_levels", ["0", "3", "16"])
def test_parallel_bvh_build_is_byte_identical(fork_levels, monkeypatch):
    """the task-parallel build forks the halves of large nodes; whatever the number of forking levels
    (0 = the sequential algorithm of reference src/bvh.cpp:35-152), the compiled tree is the oracle's, byte for
    byte — 100 353 primitives incl. a disc that dominates the root box (a lopsided first split)."""
    monkeypatch.setenv("GPUART_BVH_THREADS", fork_levels)
    descs = S.scene_d()
    q, depth = B.compile_bvh(descs)
    ref, ref_depth = O.build_bvh(descs)
    assert depth == ref_depth
    assert_bits(q, ref, "tree, fork levels " + fork_levels)


@pytest.mark.parametrize("W,H", [(640, 480), (1920, 1080), (3840, 2160), (7680, 4320), (37, 23)])
@pytest.mark.parametrize("camsel", ["default", "bench"])
def test_camera_basis_and_pixel_size(W, H, camsel):
    cam = dict(S.DEFAULT_CAMERA if camsel == "default" else S.BENCH_CAMERA)
    cam["dir"] = S.camera_dir(cam)
    a = B.camera_basis(cam["pos"], cam["dir"], cam["up"], cam["fov_y"], cam["screen_dist"], W, H)
    b = O.camera(cam["pos"], cam["dir"], cam["up"], cam["fov_y"], cam["screen_dist"], W, H)
    assert_bits(a, b, "screen basis")


def test_sun_direction():
    for az, alt in [(S.SUN_AZIMUTH, S.SUN_ALTITUDE), (0.3, 0.1), (5.0, 1.5)]:
        assert_bits(B.sun_direction(az, alt), O.sun_direction(az, alt), "sun direction")


def test_ply_loader_round_trip(tmp_path):
    """Scene D written as ASCII PLY and loaded through LoadMeshFromPLY (x10, translated) compiles to the same
    tree as the in-memory mesh."""
    model, faces = S.dragon_class_mesh(32, 24)
    path = str(tmp_path / "mesh.ply")
    S.write_ply(path, model, faces)
    disc = (S.DISC, [0, 0, 0, 0, 0, 1, 5])
    a, _, nloaded = B.compile_bvh_from_file("ply", path, 10.0, (0, 0, -0.5), [disc])
    assert nloaded == len(faces)
    w = S.load_transform(model)
    prims = [(S.TRIANGLE, row.tolist()) for row in w[faces].reshape(-1, 9)] + [disc]
    b, _ = O.build_bvh(prims)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def test_ply_loader_rejects_bad_files(tmp_path):
    p = tmp_path / "bad.ply"
    p.write_text("plx\n")
    with pytest.raises(RuntimeError):
        B.compile_bvh_from_file("ply", str(p))
    p.write_text("ply\nelement vertex 3\nelement face 1\nend_header\n0 0 0\n1 0 0\n0 1 0\n4 0 1 2 2\n")
    with pytest.raises(RuntimeError):
        B.compile_bvh_from_file("ply", str(p))  # non-triangular face
    p.write_text("ply\nelement vertex 3\nelement face 1\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n")
    with pytest.raises(RuntimeError):
        B.compile_bvh_from_file("ply", str(p))  # vertex index out of range
    with pytest.raises(RuntimeError):
        B.compile_bvh_from_file("ply", str(tmp_path / "missing.ply"))


def test_primitive_list_loader(tmp_path):
    """'sphere x y z [r]' / 'cone ...' dialect (src/utils.cpp:168-197), default radius 4, '#' comments."""
    p = tmp_path / "prims.dat"
    p.write_text("# comment\nsphere 1 2 3 0.5\nsphere -1 0 2\n\ncone 0 0 0 0 0 1 0.3 0.1\nsphere 4 4 4 1\n")
    mag, tr = 0.25, (0.0, 0.0, 1.0)
    a, _, n = B.compile_bvh_from_file("dat", str(p), mag, tr)
    assert n == 4
    f = np.float32
    def T(v): return [float(f(tr[k]) + f(mag) * f(v[k])) for k in range(3)]
    prims = [(S.SPHERE, T([1, 2, 3]) + [float(f(mag) * f(0.5))]), (S.SPHERE, T([-1, 0, 2]) + [float(f(mag) * f(4.0))]),
             (S.CONE, T([0, 0, 0]) + T([0, 0, 1]) + [float(f(mag) * f(0.3)), float(f(mag) * f(0.1))]),
             (S.SPHERE, T([4, 4, 4]) + [float(f(mag) * f(1.0))])]
    b, _ = O.build_bvh(prims)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("which", ["cluster", "tree"])
def test_cluster_and_tree_stand_ins_load_like_the_reference_scenes(tmp_path, which):
    """The synthetic stand-ins for data/cluster_100k.dat and data/tree1_21k.dat (gpuart_amd.synth_scenes), written in the
    .dat dialect and loaded as InitCluster / InitTree load the originals (src/scenes.cpp:69-103: magnification 0.01 + lift
    2.5, resp. magnification 0.3, then Disc((1,0,0),(0,0,1),6)): the C++ loader + BVH build give the tree the oracle builds
    from the descriptions synth_scenes derives with the same float32 arithmetic (these are the trees behind
    tests/golden/frames_cluster_*.npz / frames_tree_*.npz)."""
    lines = S.cluster_dat_lines(6000, 7) if which == "cluster" else S.tree_dat_lines(6, 11)
    load = S.CLUSTER_LOAD if which == "cluster" else S.TREE_LOAD
    path = str(tmp_path / (which + ".dat"))
    S.write_lines(path, lines)
    a, _, n = B.compile_bvh_from_file("dat", path, load["magnification"], load["translation"], [S.FLOOR_DISC_CT])
    descs = S.dat_descs(lines, **load) + [S.FLOOR_DISC_CT]
    assert n == len(descs) - 1
    b, _ = O.build_bvh(descs)
    assert a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


def _sort_cases():
    rng = np.random.RandomState(9)
    n = 150000
    yield "random", rng.uniform(-3, 3, n)
    yield "heavy ties", rng.randint(0, 50, n).astype(np.float64)
    yield "mesh-like ties", np.round(rng.uniform(-1, 1, n), 3)
    yield "all equal", np.zeros(n)
    yield "sorted", np.arange(n, dtype=np.float64)
    yield "reversed", np.arange(n, dtype=np.float64)[::-1]
    yield "organ pipe", np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(np.float64)
    yield "sawtooth", (np.arange(n) % 17).astype(np.float64)
    a = rng.uniform(-3, 3, n); a[rng.randint(0, n, 2000)] = np.nan; a[rng.randint(0, n, 500)] = np.inf; a[rng.randint(0, n, 500)] = -0.0
    yield "NaN, inf, -0", a
    # the classic median-of-three killer: drives introsort into its depth limit (heap sort of sub-ranges)
    m = 60000
    k = np.zeros(m)
    half = m // 2
    k[0::2][:half] = np.arange(1, half + 1)
    k[1::2][:half] = np.arange(half + 1, 2 * half + 1) if m % 2 == 0 else 0
    yield "median-of-3 killer", k
    # many tasks in the sort's pool at once (ranges above 16384 elements are tasks), half of the adjacent keys tied as in a quad mesh
    big = np.repeat(rng.uniform(-1, 1, 300000).astype(np.float32), 2)
    yield "600000 keys, every key twice", big[rng.permutation(big.size)]
    for size in (0, 1, 2, 3, 15, 16, 17, 31, 33, 100, 1000, 16385, 40000):
        yield "size %d" % size, np.round(rng.uniform(0, 8, size), 1)


def test_exact_sort_equals_std_sort():
    """csrc/host/exact_sort.h (the BVH build's parallel sort) against std::sort with the reference's comparison
    (src/bvh.cpp:96; oracle/restate/capi.cpp orc_sort_permutation): the same permutation — not just the same order of keys —
    with ties, NaNs, adversarial patterns and the depth-limit (heap sort) path, on 1, 3 and 16 threads."""
    for name, keys in _sort_cases():
        k = keys.astype(np.float32)
        want = O.sort_permutation(k)
        for threads in (1, 3, 16):
            got = B.sort_permutation(k, threads)
            assert (got == want).all(), "%s, %d threads: %d positions differ" % (name, threads, int((got != want).sum()))


def test_gather_api_rejects_bad_calls_without_a_device():
    """The multi-GPU entry points of include/gpuart_hip.h fail with GPUART_HIP_ERR_ARG (never crash) on NULL contexts and
    inconsistent shares; the share layout helpers are pure host code (a GPU box is not needed for any of this)."""
    import ctypes as C
    L = B.hip_lib()
    assert L.gpuart_hip_gather(None, 1, C.c_float(1.0), 0, None) == -1
    assert L.gpuart_hip_gather_all(None, 2, 1, C.c_float(1.0), 0, None) == -1
    assert L.gpuart_hip_comm_init(None, 2, 0, None) == -1
    assert L.gpuart_hip_comm_init_all(None, 2) == -1
    assert L.gpuart_hip_comm_destroy(None) == -1
    assert L.gpuart_hip_set_share(None, None) == -1
    g = B.share_of_rank(64, 40, 1, 3)
    assert (g.y0, g.th, g.band_rows, g.band_stride) == (8, 16, 8, 24)   # bands 1 and 4 of 5
    assert list(g.rows()) == list(range(8, 16)) + list(range(32, 40))
    empty = B.share_of_rank(64, 8, 1, 2)                                  # more ranks than bands: an empty share
    assert empty.th == 0
    full = np.full((8, 64, 4), -1.0, np.float32)
    B.scatter_rows_host(empty, np.zeros((0, 64, 4), np.float32), full)
    assert (full == -1.0).all()
    bad = B.share_of_rank(64, 40, 0, 1)
    bad.band_stride = 4                                                   # stride below the band height
    assert L.gpuart_hip_scatter_rows_host(C.byref(bad), None, None) == -1


# ---- host vector arithmetic against the reference's own compiled src/math_types.h ------------------------------------
# tests/golden/host_math.npz: inputs + outputs of oracle/_ref/libmathref.so (oracle/mathref/mathref.cpp compiled against
# /root/reference/src/math_types.h where it lies; tests/golden/make_host_golden.py). Pins rows a17 / a19 of SURVEY.md 8(a) as
# far as they are Vec3 arithmetic: both the product's host library and the oracle must reproduce it bit for bit.
def _host_math():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_math.npz"))


def test_sun_direction_equals_reference_vec3_arithmetic():
    g = _host_math()
    got_p = np.stack([B.sun_direction(az, alt) for az, alt in g["sun_in"]])
    got_o = np.stack([O.sun_direction(az, alt) for az, alt in g["sun_in"]])
    assert_bits(got_p, g["sun_out"], "product sun direction vs reference math_types.h")
    assert_bits(got_o, g["sun_out"], "oracle sun direction vs reference math_types.h")


def test_camera_basis_equals_reference_vec3_arithmetic():
    g = _host_math()
    for name, fn in (("product", B.camera_basis), ("oracle", O.camera)):
        got = np.stack([fn(c[0:3], c[3:6], c[6:9], float(c[9]), float(c[10]), int(c[11]), int(c[12])) for c in g["cam_in"]])
        assert_bits(got[:, 0:3], g["cam_in"][:, 0:3], name + " camera position")
        assert_bits(got[:, 3:13], g["cam_out"], name + " screen basis + pixel size vs reference math_types.h")


def test_vec3_operations_equal_reference_math_types():
    """Every operation of the product's Vec3<float> / Vec3<double> (csrc/host/math_types.h, part of the kept API surface) on
    the golden inputs, incl. zero vectors (NaN from normalized()), huge and tiny values."""
    g = _host_math()
    for dt, tag in ((np.float32, "vec3f"), (np.float64, "vec3d")):
        a, b, s, exp = g[tag + "_a"], g[tag + "_b"], g[tag + "_s"], g[tag + "_out"]
        got = np.stack([B.vec3_ops(a[i], b[i], s[i], dt) for i in range(len(a))])
        ui = np.uint32 if dt == np.float32 else np.uint64
        same = (got.view(ui) == exp.view(ui)) | (np.isnan(got) & np.isnan(exp))
        assert same.all(), "%s: %d of %d values differ from the reference's" % (tag, int((~same).sum()), same.size)


def test_cone_constants_equal_reference_vec3_arithmetic():
    """Cone::Cone's derived constants (src/core.cpp:191-226: axis, length, widthCoeff, cosB, dot(axis, c1), computed in double) and
    its world box, as the statements give them on the reference's own compiled Vec3f / Vec3d (oracle/mathref/mathref.cpp), for 1 024
    cones incl. cylinders, pointed and zero-length ones, |r1 - r2| at the 1e-7 cut and hostile values: the payload the oracle and the
    product store for the cone (StoreDataIntoBVH order, :230-245) equals it bit for bit; so does the node box of a one-cone tree
    wherever the cone's box holds no NaN (a NaN never enters a node box: the build's min / max comparisons skip it)."""
    g = _host_math()
    for i in range(len(g["cone_c1"])):
        d = (S.CONE, [float(x) for x in g["cone_c1"][i]] + [float(x) for x in g["cone_c2"][i]] + [float(g["cone_r1"][i]), float(g["cone_r2"][i])])
        exp = g["cone_out"][i]
        for name, fn in (("oracle", O.build_bvh), ("product", B.compile_bvh)):
            tree, _ = fn([d])
            pay = tree[4:8].ravel().astype(np.float32)
            same = (pay.view(np.uint32) == exp[:16].view(np.uint32)) | (np.isnan(pay) & np.isnan(exp[:16]))
            assert same.all(), "%s cone %d payload: got %s expected %s" % (name, i, pay, exp[:16])
            if not np.isnan(exp[16:]).any():
                box = np.concatenate([tree[0, :3], tree[1, :3]]).astype(np.float32)
                assert (box.view(np.uint32) == exp[16:].view(np.uint32)).all(), "%s cone %d box: got %s expected %s" % (name, i, box, exp[16:])


def test_uploader_sizes_the_quick_box_answers_per_tree():
    """The BVH queries answer a box from its six plane parameters where margins prove that equal to the reference's six face tests
    (csrc/hip/box_quick.h). The margins are sized per tree at upload: 4 * 2^-24 * the largest plane coordinate + 2^-90 for a tree whose
    boxes are regular, nested and free of subnormal planes — and +inf, i.e. never a quick answer, for any other
    (gpuart_hip_test_tree_slack shows the decision without a device). Every BASELINE scene must get a finite slack or the fast path is
    silently lost."""
    from gpuart_amd import binding as B
    from gpuart_amd import synth_scenes as S
    f32 = np.float32
    for name, descs in (("box", S.box_scene()), ("scene_d", S.scene_d()), ("scene_p", S.scene_p()), ("tree", S.tree_scene()), ("lattice", S.lattice_scene())):
        tree, _ = B.compile_bvh(descs)
        slack, sub = B.tree_slack(tree)
        pmax = f32(np.abs(tree[0:2, :3]).max())  # the root's box: every other box lies inside it
        assert not sub and slack == float(f32(f32(2.0 ** -22) * pmax + f32(2.0 ** -90))), (name, slack, pmax)
        assert 0 < slack < 1e-4, (name, slack)
    floor = (S.DISC, [0, 0, 0, 0, 0, 1, 6])
    inf = float("inf")
    assert B.tree_slack(B.compile_bvh([floor, (S.SPHERE, [0, 0, 1, -0.5])])[0])[0] == inf                    # a box that does not bound its sphere
    assert B.tree_slack(B.compile_bvh([(S.SPHERE, [0, 0, 1, -0.5])])[0])[0] == inf                           # an inverted box
    assert B.tree_slack(B.compile_bvh([floor, (S.TRIANGLE, [0, 0, 0.5, 1, 0, 0.5, 0, 2.0e6, 0.5])])[0])[0] == inf
    # a triangle whose lowest x is a subnormal number: regular, nested — and still no quick answers
    tiny = B.compile_bvh([floor, (S.TRIANGLE, [1.0e-40, 0, 0.5, 1, 0, 0.5, 0.5, 1, 0.5]), (S.TRIANGLE, [3, 3, 0.5, 4, 3, 0.5, 3, 4, 0.5])])[0]
    c = B.tree_class(tiny)
    assert not c["irregular"] and not c["disorderly"]
    assert B.tree_slack(tiny) == (inf, True)
    # ... while the same scene with that coordinate at 0 takes them
    zero = B.compile_bvh([floor, (S.TRIANGLE, [0.0, 0, 0.5, 1, 0, 0.5, 0.5, 1, 0.5]), (S.TRIANGLE, [3, 3, 0.5, 4, 3, 0.5, 3, 4, 0.5])])[0]
    assert B.tree_slack(zero)[0] < 1e-5 and not B.tree_slack(zero)[1]


def test_uploader_classifies_trees_for_the_visiting_order():
    """The fast kernels visit a node's nearer child first only where boxes bound what they hold ("Nearer child
    first, with a certificate"); gpuart_hip_upload_bvh decides per tree, and gpuart_hip_test_tree_class shows the decision without a
    device. Every scene of BASELINE.json and of the reference's scene list must come out order-free (or the fast path is silently
    lost); what the hostile classes produce must not."""
    from gpuart_amd import binding as B
    from gpuart_amd import synth_scenes as S
    T = lambda descs: B.tree_class(B.compile_bvh(descs)[0])
    for name, descs, mask in (("box", S.box_scene(), 15), ("scene_d", S.scene_d(), 6), ("scene_p", S.scene_p(), 3), ("tree", S.tree_scene(), 11),
                              ("lattice", S.lattice_scene(), 7), ("one sphere of radius 0", [(S.SPHERE, [0, 0, 1, 0])], 1)):
        c = T(descs)
        assert c == dict(irregular=False, disorderly=False, type_mask=mask), (name, c)
    flagged = 0
    for seed in range(200):  # the plain random class (degenerate but honest primitives): only its empty scenes are held back
        prims = S.random_case(seed)["prims"]
        c = T(prims)
        assert not c["disorderly"], seed
        assert c["irregular"] == (len(prims) == 0), seed
        assert not T(S.random_lattice_case(seed)["prims"])["disorderly"], seed
        flagged += c["irregular"]
    assert 0 < flagged < 20
    floor = (S.DISC, [0, 0, 0, 0, 0, 1, 6])
    # a box that does not bound its primitive in any useful sense, each for its own reason
    assert T([floor, (S.SPHERE, [0, 0, 1, -0.5])])["disorderly"]                                  # negative radius (its leaf's box is saved by the floor)
    assert T([(S.SPHERE, [0, 0, 1, -0.5])])["irregular"]                                          # alone: an inverted box
    assert T([floor, (S.TRIANGLE, [0, 0, -1e30, 1, 0, 0, 0, 1, 0])])["disorderly"]                # a vertex that swallows the ray origin
    assert T([floor, (S.TRIANGLE, [0, 0, 0.5, 1, 0, 0.5, 0, 2.0e6, 0.5])])["disorderly"]          # beyond 2^20
    assert T([floor, (S.SPHERE, [0, 0, 1, 2.0e6])])["disorderly"]
    assert not T([floor, (S.TRIANGLE, [0, 0, 0.5, 1, 0, 0.5, 0, 1.0e6, 0.5])])["disorderly"]      # large but within bounds
    for bad in (float("nan"), float("inf")):
        c = T([floor, (S.SPHERE, [0, bad, 1, 0.5])])
        assert c["irregular"] or c["disorderly"]
    c = T([floor, (S.CONE, [0, 0, 0, 0, 0, 1, 0.2, -0.1])])                                       # a cone with a negative radius
    assert c["irregular"] or c["disorderly"]
    # a hand-made tree: a child pushed outside its parent, a cone whose axis contradicts its centres
    tree, _ = B.compile_bvh(S.scene_p())
    assert not B.tree_class(tree)["disorderly"]
    hacked = tree.copy(); hacked[3, 0] -= 100.0   # the lower child of the root: bbmin.x far outside
    assert B.tree_class(hacked) == dict(irregular=False, disorderly=True, type_mask=3)
    ctree, _ = B.compile_bvh([floor, (S.CONE, [0, 0, 0.2, 0.3, 0.1, 0.9, 0.2, 0.1])])
    assert not B.tree_class(ctree)["disorderly"]
    rows = np.nonzero(ctree[:, 0].view(np.uint32) == S.CONE)[0]
    k = int([r for r in rows if ctree[r, 1] == 0 and ctree[r, 2] == 0][0])  # the cone's type quad; its data quads follow
    bent = ctree.copy(); bent[k + 3, 0:3] = (1.0, 0.0, 0.0)                # axis no longer points from centre 1 to centre 2
    assert B.tree_class(bent)["disorderly"]
    # ... or whose derived constants lie about it: dotAxC1 positions the surface cone_hit intersects, cosB its normal, the axis must be a unit vector
    # (the data quads: {c1, r1}{c2, r2}{axis, len}{widthCoeff, cosB, dotAxC1, pad}, reference src/core.cpp:230-245)
    for col, delta in ((2, 0.05), (1, 0.05), (0, 0.05)):
        lied = ctree.copy(); lied[k + 4, col] += delta
        assert B.tree_class(lied)["disorderly"], "cone constant %d off by %g went unnoticed" % (col, delta)
    scaled = ctree.copy(); scaled[k + 3, 0:3] *= 1.5; scaled[k + 3, 3] /= 1.5   # axis x 1.5, length / 1.5: centre 2 still agrees, the axis is no unit vector
    assert B.tree_class(scaled)["disorderly"]
    for seed in range(40):   # the honest cones of the random classes stay order-free (the check compares with the reference's own double-precision formulas)
        case = S.random_case(seed)["prims"]
        if any(t == S.CONE for t, _ in case):
            assert not T(case)["disorderly"], seed
    with pytest.raises(B.HipError, match="malformed"):
        B.tree_class(tree[:5])
