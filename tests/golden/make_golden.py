#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the reference's UNMODIFIED GLSL on Mesa llvmpipe.

Container-only (needs /root/reference/shaders and oracle/_ref/libglref.so — `make -C oracle`).
The fixtures are data: seeded inputs + the outputs the reference shaders produced for them.
No reference source text is stored. Usage:  python tests/golden/make_golden.py [section ...]

Per-function vectors come from tiny probe main()s of our own, linked against the unmodified
reference shader objects (the reference links separately compiled objects the same way,
src/renderer.cpp:259-359). Frames come from the reference's four programs linked from the
same object lists as src/renderer.cpp:259-359, driven with the uniforms of
src/renderer.cpp:372-404,534-599. Host-side inputs (compiled BVH, camera basis, Sun direction,
RandSeed, PixelSize) come from the oracle's host restatement (the reference's C++ host files
need the un-vendored nanogui headers and are not buildable here — DESIGN.md "Oracle").
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
import glref  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

D_RANDOM = "float random(float x); float random(vec2 x); float random(vec3 x); float random(vec4 x);"
D_SPHERE = "void SphereIntersection(in vec3 a, in vec3 b, in vec3 c, in float r, out float pos, out vec3 p, out vec3 n);"
D_DISC = "void DiscIntersection(in vec3 a, in vec3 b, in vec3 c, in float r, in vec3 dn, out float pos, out vec3 p, out vec3 n);"
D_TRI = ("void TriangleIntersection(in vec3 a, in vec3 b, in vec3 v0, in vec3 v1, in vec3 v2, out float pos, "
         "out vec3 p, out vec3 n, out vec2 uv);")
D_CONE = ("void ConeIntersection(in vec3 a, in vec3 b, in vec4 q0, in vec4 q1, in vec4 q2, in float w, in float cb, "
          "in float dc, out float pos, out vec3 p, out vec3 n);")
D_AABB = "bool IntersectsAABB(in vec3 rs, in vec3 rd, in vec3 rdiv, in samplerBuffer t, in int addr, out float pos);"
D_BVH = "void CheckBVHIntersection(in vec3 rs, in vec3 rd, in samplerBuffer t, out float pos, out vec3 p, out vec3 n, out int ptype);"
D_INCL = ("void CheckIntersectionInclUserSphere(in vec3 rs, in vec3 rd, in samplerBuffer t, in vec4 us, out float pos, "
          "out vec3 p, out vec3 n, out int ptype, out bool ush);")
D_HEMI = "vec3 GetRandomHemisphereDirection(in vec3 v, in vec3 ri);"
D_ICONE = "vec3 GetRandomDirectionInsideCone(in vec3 v, in vec3 n, in float ha, in vec3 ri);"
D_SKY = "vec3 GetSkyColor(in vec3 dir, in vec4 sda);"

GEOM = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl"]
HIT_OUT = "if (pos > 0) { O0 = vec4(pos, p); O1 = vec4(n, 0); } else { O0 = vec4(pos, 0, 0, 0); O1 = vec4(0); }"


def pad4(a, w=0.0):
    a = np.asarray(a, np.float32)
    if a.shape[1] == 4:
        return a
    return np.concatenate([a, np.full((a.shape[0], 4 - a.shape[1]), w, np.float32)], 1)


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------
def gen_hash(gl):
    rng = np.random.RandomState(11)
    x = rng.uniform(-4, 4, (4096, 4)).astype(np.float32)
    x[:8] = [[0, 0, 0, 0], [1, 2, 3, 4], [-0.0, 0, 0, 0], [1e-30, 1e30, -1e-30, -1e30], [0.5, 0.5, 0.5, 0.5],
             [0.81472367, 0.135477006, 0.905791938, 0.835008562], [3, 3, 3, 3], [-1, -2, -3, -4]]
    o, = glref.run_probe(gl, "O0 = vec4(random(i0.x), random(i0.xy), random(i0.xyz), random(i0));",
                         ["noise.glsl"], [x], 1, decls=D_RANDOM)
    save("hash", x=x, out=o)


def gen_llvmpipe_math(gl):
    """sin/cos/pow(x,16) of the GL driver itself — pins the oracle's transcendental emulation."""
    rng = np.random.RandomState(12)
    a = np.concatenate([rng.uniform(0, 6.2831855, 10000), rng.uniform(-8, 8, 4000), rng.uniform(-1e-3, 1e-3, 1000),
                        np.linspace(0, 6.2831852, 1384)]).astype(np.float32)
    w = np.concatenate([rng.uniform(0, 1, 14000), 1 - rng.uniform(0, 1e-3, 1000), np.linspace(0, 1, 1384)]).astype(np.float32)
    x = np.zeros((16384, 4), np.float32)
    x[:, 0] = a
    x[:, 1] = w
    o, = glref.run_probe_big(gl, "O0 = vec4(sin(i0.x), cos(i0.x), pow(i0.y, 16), sqrt(i0.y));", [], [x], 1)
    save("llvmpipe_math", x=x[:, :2].copy(), out=o)


def gen_hemisphere(gl):
    rng = np.random.RandomState(13)
    n = 4096
    v = unit(rng.normal(size=(n, 3))).astype(np.float32)
    v[:64] = [0, 0, 1]
    v[64:128] = [0, 0, -1]
    v[128:160] = unit(np.array([[1e-7, 5e-7, 1.0]])).astype(np.float32)
    ri = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    o, = glref.run_probe(gl, "O0 = vec4(GetRandomHemisphereDirection(i0.xyz, i1.xyz), 0);", ["common.glsl", "noise.glsl"],
                         [pad4(v), pad4(ri)], 1, decls=D_HEMI)
    save("hemisphere", v=v, ri=ri, out=o[:, :3].copy())
    # fuzzy specular cone sampler (common.glsl:81-106): v = reflected dir (un-normalised), normal unit
    nrm = unit(rng.normal(size=(n, 3))).astype(np.float32)
    vv = (unit(rng.normal(size=(n, 3)) + 1.5 * nrm) * rng.uniform(0.2, 3, (n, 1))).astype(np.float32)
    o, = glref.run_probe(gl, "O0 = vec4(GetRandomDirectionInsideCone(i0.xyz, i1.xyz, 10 * 3.14159/180, i2.xyz), 0);",
                         ["common.glsl", "noise.glsl"], [pad4(vv), pad4(nrm), pad4(ri)], 1, decls=D_ICONE)
    save("inside_cone", v=vv, normal=nrm, ri=ri, out=o[:, :3].copy())


def rays_towards(rng, n, target, spread):
    rs = rng.uniform(-3, 3, (n, 3))
    aim = target + rng.normal(size=(n, 3)) * spread
    rd = (aim - rs) * rng.uniform(0.05, 2.0, (n, 1))
    return rs.astype(np.float32), rd.astype(np.float32)


def gen_sphere(gl):
    rng = np.random.RandomState(14)
    n = 4096
    c = rng.uniform(-1, 1, (n, 3))
    r = rng.uniform(0.05, 1.0, (n, 1))
    rs, rd = rays_towards(rng, n, c, r * 0.8)
    sph = np.concatenate([c, r], 1).astype(np.float32)
    sph[:256, 3] = 0  # radius 0 = the disabled user sphere (src/main.cpp:622)
    rs[256:512] = (sph[256:512, :3] + 0.3 * sph[256:512, 3:4] * unit(rng.normal(size=(256, 3)))).astype(np.float32)  # inside
    o = glref.run_probe(gl, "float pos; vec3 p, n; SphereIntersection(i0.xyz, i1.xyz, i2.xyz, i2.w, pos, p, n);" + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), sph], 2, decls=D_SPHERE)
    save("sphere", rs=rs, rd=rd, sph=sph, o0=o[0], o1=o[1])


def gen_disc(gl):
    rng = np.random.RandomState(15)
    n = 4096
    c = rng.uniform(-1, 1, (n, 3))
    r = rng.uniform(0.05, 1.5, (n, 1))
    dn = unit(rng.normal(size=(n, 3))).astype(np.float32)
    dn[:512] = [0, 0, 1]
    rs, rd = rays_towards(rng, n, c, r * 0.9)
    rd[512:640, 2] = 0  # parallel to the z=const discs? only where dn=(0,0,1): rows <512 unaffected
    rd[:64, 2] = 0      # grazing: dot(rdir, n) == 0
    cr = np.concatenate([c, r], 1).astype(np.float32)
    o = glref.run_probe(gl, "float pos; vec3 p, n; DiscIntersection(i0.xyz, i1.xyz, i2.xyz, i2.w, i3.xyz, pos, p, n);" + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), cr, pad4(dn)], 2, decls=D_DISC)
    save("disc", rs=rs, rd=rd, cr=cr, dn=dn, o0=o[0], o1=o[1])


def gen_triangle(gl):
    rng = np.random.RandomState(16)
    n = 4096
    v0 = rng.uniform(-1, 1, (n, 3))
    v1 = v0 + rng.normal(size=(n, 3)) * rng.uniform(0.01, 1, (n, 1))
    v2 = v0 + rng.normal(size=(n, 3)) * rng.uniform(0.01, 1, (n, 1))
    bary = rng.dirichlet([1, 1, 1], n)
    inside = bary[:, :1] * v0 + bary[:, 1:2] * v1 + bary[:, 2:] * v2
    cen = (v0 + v1 + v2) / 3
    aim = np.where(rng.uniform(size=(n, 1)) < 0.6, inside, cen + (inside - cen) * 2.5)
    rs = rng.uniform(-3, 3, (n, 3))
    rd = (aim - rs) * rng.uniform(0.05, 2.0, (n, 1))
    # edge/vertex grazing: aim exactly at a vertex / edge midpoint
    aim[:128] = v0[:128]
    aim[128:256] = (v1[128:256] + v2[128:256]) / 2
    rd[:256] = (aim[:256] - rs[:256])
    # degenerate triangles
    v2[256:288] = v1[256:288]
    f = lambda a: a.astype(np.float32)
    rs, rd, v0, v1, v2 = f(rs), f(rd), f(v0), f(v1), f(v2)
    o = glref.run_probe(gl, "float pos; vec3 p, n; vec2 uv; TriangleIntersection(i0.xyz, i1.xyz, i2.xyz, i3.xyz, i4.xyz, pos, p, n, uv);"
                        + HIT_OUT, GEOM, [pad4(rs), pad4(rd), pad4(v0), pad4(v1), pad4(v2)], 2, decls=D_TRI)
    save("triangle", rs=rs, rd=rd, v0=v0, v1=v1, v2=v2, o0=o[0], o1=o[1])


def cone_quads(c1, c2, r1, r2):
    """StoreDataIntoBVH payload of cones via the oracle's host restatement (16 floats each)."""
    descs = [(S.CONE, list(map(float, c1[i])) + list(map(float, c2[i])) + [float(r1[i]), float(r2[i])]) for i in range(len(c1))]
    q = np.zeros((len(descs), 16), np.float32)
    for i, d in enumerate(descs):
        tree, _ = O.build_bvh([d])
        assert tree.shape[0] == 3 + 1 + 4
        q[i] = tree[4:8].ravel()
    return q


def gen_cone(gl):
    rng = np.random.RandomState(17)
    n = 4096
    c1 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    ax = unit(rng.normal(size=(n, 3)))
    ln = rng.uniform(0.1, 1.5, (n, 1))
    c2 = (c1 + ax * ln).astype(np.float32)
    r1 = rng.uniform(0.02, 0.5, n).astype(np.float32)
    r2 = rng.uniform(0.02, 0.5, n).astype(np.float32)
    r2[:512] = r1[:512]          # cylinders (CosB = 0)
    r2[512:640] = 0.0            # pointed
    q = cone_quads(c1, c2, r1, r2)
    mid = (c1 + c2) / 2 + ax * ln * rng.uniform(-0.6, 0.6, (n, 1))
    rs, rd = rays_towards(rng, n, mid, np.maximum(r1, r2)[:, None] * 0.9)
    rs[640:768] = (mid[640:768] + 0.2 * np.minimum(r1, r2)[640:768, None] * unit(rng.normal(size=(128, 3)))).astype(np.float32)
    o = glref.run_probe(gl, "float pos; vec3 p, n; ConeIntersection(i0.xyz, i1.xyz, i2, i3, i4, i5.x, i5.y, i5.z, pos, p, n);"
                        "if (pos < 1.0e-4) pos = -1;" + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), q[:, 0:4], q[:, 4:8], q[:, 8:12], q[:, 12:16]], 2, decls=D_CONE)
    save("cone", rs=rs, rd=rd, c1=c1, c2=c2, r1=r1, r2=r2, quads=q, o0=o[0], o1=o[1])


HOSTILE = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-30, -1e-30, -0.0, 0.0, 1e19, -1e19, 1e-4, -1e-4, 1e-8, 1e-10, 3.4028234e38,
                    -3.4028234e38, 1.17549435e-38, 1e-39, 1e-45, -1e-45, 1e10, -1e10, 0.5, -0.3, -1.0, 1.0], np.float32)
CUT = "if (pos < 1.0e-4) pos = -1;"  # CheckBVHPrimitiveIntersection's visibility cut (bvh_intersection.glsl:170), as in the cone probe


def inject(rng, arrays, frac=0.35):
    """Replaces 1-3 random components of `frac` of the rows (of the arrays taken together) by hostile values or by their negation."""
    n = len(arrays[0])
    widths = [a.shape[1] for a in arrays]
    for i in np.nonzero(rng.uniform(size=n) < frac)[0]:
        for _ in range(int(rng.randint(1, 4))):
            k = int(rng.randint(len(arrays)))
            c = int(rng.randint(widths[k]))
            arrays[k][i, c] = -arrays[k][i, c] if rng.uniform() < 0.2 else HOSTILE[int(rng.randint(len(HOSTILE)))]


def gen_intersect_wild(gl):
    """The four intersectors on hostile numbers (NaN, +-inf, +-1e30, denormals, negative radii, the reference's own magic numbers)
    in the primitive AND in the ray: what wild scenes feed them. Outputs as CheckBVHPrimitiveIntersection leaves them."""
    n = 4096
    f = lambda a: np.ascontiguousarray(a, np.float32)
    rng = np.random.RandomState(9001)
    c = rng.uniform(-1, 1, (n, 3)); r = rng.uniform(0.05, 1.0, (n, 1))
    rs, rd = rays_towards(rng, n, c, r * 0.8)
    rs, rd, sph = f(rs), f(rd), f(np.concatenate([c, r], 1))
    inject(rng, [rs, rd, sph])
    o = glref.run_probe(gl, "float pos; vec3 p, n; SphereIntersection(i0.xyz, i1.xyz, i2.xyz, i2.w, pos, p, n);" + CUT + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), sph], 2, decls=D_SPHERE)
    save("sphere_wild", rs=rs, rd=rd, sph=sph, o0=o[0], o1=o[1])

    rng = np.random.RandomState(9002)
    c = rng.uniform(-1, 1, (n, 3)); r = rng.uniform(0.05, 1.5, (n, 1)); dn = unit(rng.normal(size=(n, 3)))
    rs, rd = rays_towards(rng, n, c, r * 0.9)
    rs, rd, cr, dn = f(rs), f(rd), f(np.concatenate([c, r], 1)), f(dn)
    inject(rng, [rs, rd, cr, dn])
    o = glref.run_probe(gl, "float pos; vec3 p, n; DiscIntersection(i0.xyz, i1.xyz, i2.xyz, i2.w, i3.xyz, pos, p, n);" + CUT + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), cr, pad4(dn)], 2, decls=D_DISC)
    save("disc_wild", rs=rs, rd=rd, cr=cr, dn=dn, o0=o[0], o1=o[1])

    rng = np.random.RandomState(9003)
    v0 = rng.uniform(-1, 1, (n, 3))
    v1 = v0 + rng.normal(size=(n, 3)) * rng.uniform(0.01, 1, (n, 1)); v2 = v0 + rng.normal(size=(n, 3)) * rng.uniform(0.01, 1, (n, 1))
    bary = rng.dirichlet([1, 1, 1], n)
    aim = bary[:, :1] * v0 + bary[:, 1:2] * v1 + bary[:, 2:] * v2
    rs = rng.uniform(-3, 3, (n, 3)); rd = (aim - rs) * rng.uniform(0.05, 2.0, (n, 1))
    rs, rd, v0, v1, v2 = f(rs), f(rd), f(v0), f(v1), f(v2)
    inject(rng, [rs, rd, v0, v1, v2])
    o = glref.run_probe(gl, "float pos; vec3 p, n; vec2 uv; TriangleIntersection(i0.xyz, i1.xyz, i2.xyz, i3.xyz, i4.xyz, pos, p, n, uv);"
                        + CUT + HIT_OUT, GEOM, [pad4(rs), pad4(rd), pad4(v0), pad4(v1), pad4(v2)], 2, decls=D_TRI)
    save("triangle_wild", rs=rs, rd=rd, v0=v0, v1=v1, v2=v2, o0=o[0], o1=o[1])

    rng = np.random.RandomState(9004)
    c1 = f(rng.uniform(-1, 1, (n, 3))); ax = unit(rng.normal(size=(n, 3))); ln = rng.uniform(0.1, 1.5, (n, 1))
    c2 = f(c1 + ax * ln); r1 = f(rng.uniform(0.02, 0.5, (n, 1))); r2 = f(rng.uniform(0.02, 0.5, (n, 1)))
    mid = (c1 + c2) / 2 + ax * ln * rng.uniform(-0.6, 0.6, (n, 1))
    rs, rd = rays_towards(rng, n, mid, np.maximum(r1, r2) * 0.9)
    rs, rd = f(rs), f(rd)
    half = n // 2
    inject(rng, [c1[:half], c2[:half], r1[:half], r2[:half]], 0.5)       # hostile cone parameters through Cone::Cone's double arithmetic
    q = cone_quads(c1, c2, r1[:, 0], r2[:, 0])
    inject(rng, [rs, rd])
    inject(rng, [q[half:]], 0.5)                                         # hostile values straight in the payload the shader reads
    o = glref.run_probe(gl, "float pos; vec3 p, n; ConeIntersection(i0.xyz, i1.xyz, i2, i3, i4, i5.x, i5.y, i5.z, pos, p, n);" + CUT + HIT_OUT,
                        GEOM, [pad4(rs), pad4(rd), q[:, 0:4], q[:, 4:8], q[:, 8:12], q[:, 12:16]], 2, decls=D_CONE)
    save("cone_wild", rs=rs, rd=rd, c1=c1, c2=c2, r1=r1, r2=r2, quads=q, o0=o[0], o1=o[1])


def gen_shade_wild(gl):
    """random(), the two direction samplers and the sky on hostile inputs (what a path that bounced off a wild primitive carries:
    NaN / infinite hit points and normals, huge seeds)."""
    n = 4096
    rng = np.random.RandomState(9101)
    x = rng.uniform(-4, 4, (n, 4)).astype(np.float32)
    inject(rng, [x], 0.6)
    o, = glref.run_probe(gl, "O0 = vec4(random(i0.x), random(i0.xy), random(i0.xyz), random(i0));", ["noise.glsl"], [x], 1, decls=D_RANDOM)
    save("hash_wild", x=x, out=o)
    v = unit(rng.normal(size=(n, 3))).astype(np.float32)
    ri = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    inject(rng, [v, ri], 0.5)
    o, = glref.run_probe(gl, "O0 = vec4(GetRandomHemisphereDirection(i0.xyz, i1.xyz), 0);", ["common.glsl", "noise.glsl"],
                         [pad4(v), pad4(ri)], 1, decls=D_HEMI)
    save("hemisphere_wild", v=v, ri=ri, out=o[:, :3].copy())
    nrm = unit(rng.normal(size=(n, 3))).astype(np.float32)
    vv = (unit(rng.normal(size=(n, 3)) + 1.5 * nrm) * rng.uniform(0.2, 3, (n, 1))).astype(np.float32)
    ri2 = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    inject(rng, [vv, nrm, ri2], 0.5)
    o, = glref.run_probe(gl, "O0 = vec4(GetRandomDirectionInsideCone(i0.xyz, i1.xyz, 10 * 3.14159/180, i2.xyz), 0);",
                         ["common.glsl", "noise.glsl"], [pad4(vv), pad4(nrm), pad4(ri2)], 1, decls=D_ICONE)
    save("inside_cone_wild", v=vv, normal=nrm, ri=ri2, out=o[:, :3].copy())
    d = (unit(rng.normal(size=(n, 3))) * rng.uniform(0.1, 3, (n, 1))).astype(np.float32)
    inject(rng, [d], 0.5)
    alt = np.float32(0.7)
    sun = O.sun_direction(S.SUN_AZIMUTH, alt)
    sda = [float(sun[0]), float(sun[1]), float(sun[2]), float(alt)]
    o, = glref.run_probe(gl, "O0 = vec4(GetSkyColor(i0.xyz, i1), 0);", ["sky.glsl"], [pad4(d), np.tile(np.array(sda, np.float32), (n, 1))], 1, decls=D_SKY)
    save("sky_wild", dir=d, sun_dir_alt=np.array(sda, np.float32), out=o[:, :3].copy())


def gen_aabb(gl):
    rng = np.random.RandomState(18)
    n = 4096
    lo = rng.uniform(-2, 2, (n, 3))
    hi = lo + rng.uniform(0, 1.5, (n, 3)) * (rng.uniform(size=(n, 3)) > 0.1)  # some flat boxes
    cen = (lo + hi) / 2
    rs, rd = rays_towards(rng, n, cen, (hi - lo) * 0.7 + 0.02)
    rs[:512] = (lo[:512] + (hi[:512] - lo[:512]) * rng.uniform(0, 1, (512, 3))).astype(np.float32)  # origin inside
    rd[512:768, 0] = 0
    rd[768:1024, 1] = 0
    rd[1024:1280, 2] = 0
    rs[1280:1408, 0] = lo[1280:1408, 0].astype(np.float32)  # origin on a face plane
    lo, hi = lo.astype(np.float32), hi.astype(np.float32)
    boxes = np.zeros((2 * n, 4), np.float32)
    boxes[0::2, :3] = lo
    boxes[1::2, :3] = hi
    outs = []
    chunk = 4096
    for s in range(0, n, chunk):
        o, = glref.run_probe(gl, "float pos; bool h = IntersectsAABB(i0.xyz, i1.xyz, 1/i1.xyz, BVH, 2*int(gl_FragCoord.x), pos);"
                             "O0 = vec4(h ? 1.0 : 0.0, h ? pos : 0.0, 0, 0);",
                             GEOM + ["bvh_intersection.glsl"], [pad4(rs[s:s + chunk]), pad4(rd[s:s + chunk])], 1,
                             bvh=boxes[2 * s:2 * (s + chunk)], decls=D_AABB)
        outs.append(o)
    save("aabb", rs=rs, rd=rd, bmin=lo, bmax=hi, out=np.concatenate(outs)[:, :2].copy())


def gen_aabb_irregular(gl):
    """IntersectsAABB on boxes the build can produce from wild primitives: one axis inverted (a sphere / disc with a negative
    or infinite radius), NaN or infinite bounds, and rays with zero, infinite or NaN components. The reference can hit a box
    with ONE irregular axis through that axis's two planes (those faces check only the other two axes), and a plane at +-inf
    counts as an intersection that leaves the entry at 1e19."""
    rng = np.random.RandomState(4242)
    n = 4096
    inf, nan = np.float32(np.inf), np.float32(np.nan)
    lo = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    hi = (lo + rng.uniform(0, 1.5, (n, 3))).astype(np.float32)
    cen = (lo + hi) / 2
    rs, rd = rays_towards(rng, n, cen, (hi - lo) * 0.7 + 0.02)
    ax = rng.randint(0, 3, n)
    I = np.arange(n)
    k = I % 16
    # one irregular axis, by kind
    sw = k == 0; lo[sw, ax[sw]], hi[sw, ax[sw]] = hi[sw, ax[sw]].copy(), lo[sw, ax[sw]].copy()      # inverted, finite
    m = k == 1; lo[m, ax[m]] = np.float32(9.9e30); hi[m, ax[m]] = np.float32(-9.9e30)               # the build's initial values
    m = k == 2; lo[m, ax[m]] = inf; hi[m, ax[m]] = -inf                                             # c - (-inf), c + (-inf)
    m = k == 3; lo[m, ax[m]] = nan
    m = k == 4; hi[m, ax[m]] = nan
    m = k == 5; lo[m, ax[m]] = nan; hi[m, ax[m]] = nan
    m = k == 6; lo[m, ax[m]] = -inf                                                                 # regular but infinite
    m = k == 7; hi[m, ax[m]] = inf
    m = k == 8; lo[m, ax[m]] = -inf; hi[m, ax[m]] = inf
    m = k == 9; lo[m] = -inf; hi[m] = inf; hi[m, ax[m]] = lo[m, ax[m]] = np.float32(1.5)            # an infinite plate
    m = k == 10; lo[m] = -inf; hi[m] = inf; lo[m, ax[m]] = np.float32(9.9e30); hi[m, ax[m]] = np.float32(-9.9e30)  # the box of seed 42874's leaf
    m = k == 11; lo[m] = -inf; hi[m] = -inf                                                         # a point at -inf
    m = k == 12; a2 = (ax[m] + 1) % 3; lo[m, ax[m]] = inf; hi[m, ax[m]] = -inf; lo[m, a2] = nan     # two irregular axes: never hit
    m = k == 13; lo[m, ax[m]] = -inf; hi[m, ax[m]] = -inf
    m = k == 14; lo[m, ax[m]] = np.float32(1e30); hi[m, ax[m]] = inf
    # k == 15: regular boxes, wild rays only
    j = (I // 16) % 8
    m = j == 1; rd[m, ax[m]] = 0
    m = j == 2; rd[m, (ax[m] + 1) % 3] = 0
    m = j == 3; rs[m, ax[m]] = inf
    m = j == 4; rs[m, (ax[m] + 2) % 3] = nan
    m = j == 5; rd[m, ax[m]] = inf
    m = j == 6; rd[m, ax[m]] = np.float32(1e30)
    m = j == 7; rs[m, ax[m]] = np.float32(-1e30)
    boxes = np.zeros((2 * n, 4), np.float32)
    boxes[0::2, :3] = lo
    boxes[1::2, :3] = hi
    o, = glref.run_probe(gl, "float pos; bool h = IntersectsAABB(i0.xyz, i1.xyz, 1/i1.xyz, BVH, 2*int(gl_FragCoord.x), pos);"
                         "O0 = vec4(h ? 1.0 : 0.0, h ? pos : 0.0, 0, 0);",
                         GEOM + ["bvh_intersection.glsl"], [pad4(rs), pad4(rd)], 1, bvh=boxes, decls=D_AABB)
    print("aabb_irregular: %d of %d boxes hit" % (int((o[:, 0] > 0).sum()), n))
    save("aabb_irregular", rs=rs, rd=rd, bmin=lo, bmax=hi, out=o[:, :2].copy())


def gen_sky(gl):
    rng = np.random.RandomState(19)
    n = 4096
    d = (unit(rng.normal(size=(n, 3))) * rng.uniform(0.1, 3, (n, 1))).astype(np.float32)
    d[:64, 2] = np.abs(d[:64, 2]) * 1e-3   # near horizon
    d[64:128, :2] *= 1e-3                  # near zenith
    for k, alt in enumerate([np.float32(3.1415926) / np.float32(4), np.float32(0.1), np.float32(1.5)]):
        sun = O.sun_direction(S.SUN_AZIMUTH, alt)
        sda = [float(sun[0]), float(sun[1]), float(sun[2]), float(alt)]
        o, = glref.run_probe(gl, "O0 = vec4(GetSkyColor(i0.xyz, i1), 0);", ["sky.glsl"],
                             [pad4(d), np.tile(np.array(sda, np.float32), (n, 1))], 1, decls=D_SKY)
        save("sky_%d" % k, dir=d, sun_dir_alt=np.array(sda, np.float32), out=o[:, :3].copy())


def gen_uv(gl):
    """Interpolated UV of the full-screen quad (vertex.glsl:29-37) as llvmpipe rasterises it."""
    fs = gl.shader_src(glref.GL_FRAGMENT_SHADER, "#version 330 core\nin vec2 UV; layout(location=0) out vec4 O0;"
                       "void main(){ O0 = vec4(UV, gl_FragCoord.xy); }")
    prog = gl.program([gl.ref_shader("vertex.glsl"), fs])
    rng = np.random.RandomState(20)
    out = {}
    for W, H in [(1, 1), (2, 3), (7, 5), (37, 23), (64, 36), (96, 96), (100, 100), (128, 72), (640, 480),
                 (1920, 1080), (3840, 2160), (7680, 4320)]:
        t = gl.tex(W, H)
        fb = gl.fbo([t])
        gl.draw(prog, fb, W, H)
        gl.finish()
        o = gl.read(t, W, H)
        gl.L.glref_delete_fbo(fb)
        gl.L.glref_delete_tex(t)
        assert (o[:, :, 2] == np.arange(W)[None, :] + 0.5).all() and (o[:, :, 3] == np.arange(H)[:, None] + 0.5).all()
        if W * H <= 10000:
            out["full_%dx%d" % (W, H)] = o[:, :, :2].copy()
        else:
            n = 4096
            xs = rng.randint(0, W, n); ys = rng.randint(0, H, n)
            d = np.arange(min(W, H))  # pixels hugging the diagonal
            xs[:1024] = (d[:: max(1, len(d) // 1024)][:1024] * W // min(W, H))[:1024] if len(d) >= 1024 else xs[:1024]
            ys[:1024] = np.clip((xs[:1024] * H) // W + rng.randint(-1, 2, 1024), 0, H - 1)
            out["xy_%dx%d" % (W, H)] = np.stack([xs, ys], 1).astype(np.int32)
            out["uv_%dx%d" % (W, H)] = o[ys, xs, :2].copy()
    save("uv", **out)


# ---- scenes / frames ---------------------------------------------------------------------------
class RefPrograms:
    """The reference's four programs, linked from the object lists of src/renderer.cpp:259-359."""

    def __init__(self, gl, max_segments=None):
        self.gl = gl
        geo = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "intersection.glsl", "sky.glsl",
               "bvh_intersection.glsl"]
        r = gl.ref_shader
        self.cam = gl.program([r("cam_init.glsl"), r("vertex.glsl")])
        self.direct = gl.program([r(x) for x in geo] + [r("direct_lighting.glsl"), r("common.glsl"), r("vertex.glsl")])
        edit = None
        if max_segments is not None and max_segments != 5:
            edit = ("const int MAX_PATH_SEGMENTS = 5;", "const int MAX_PATH_SEGMENTS = %d;" % max_segments)
        self.pt = gl.program([r(x) for x in geo] + [r("path_tracing.glsl", edit=edit), r("common.glsl"), r("noise.glsl"),
                                                    r("vertex.glsl")])
        self.norm = gl.program([r("pt_normalize.glsl"), r("vertex.glsl")])


class RefRenderer:
    """Drives the reference programs the way src/renderer.cpp does (uniform for uniform)."""

    def __init__(self, gl, progs, W, H, cam, tree):
        self.gl, self.p, self.W, self.H = gl, progs, W, H
        self.camd = cam
        self.cam = O.camera(cam["pos"], cam["dir"], cam["up"], cam["fov_y"], cam["screen_dist"], W, H)
        self.tbo = gl.tbo(tree)
        self.rstart, self.rdir = gl.tex(W, H), gl.tex(W, H)
        fb = gl.fbo([self.rstart, self.rdir])
        c = self.cam
        gl.set_uniforms(progs.cam, Pos=c[0:3], BottomLeft=c[3:6], DeltaHorz=c[6:9], DeltaVert=c[9:12])
        gl.draw(progs.cam, fb, W, H)
        self.acc = [gl.tex(W, H), gl.tex(W, H)]
        self.accfb = [gl.fbo([t]) for t in self.acc]
        self.out = gl.tex(W, H)
        self.outfb = gl.fbo([self.out])
        self.sun_az, self.sun_alt, self.sun_on = S.SUN_AZIMUTH, S.SUN_ALTITUDE, 1
        self.us, self.us_em, self.us_flags = S.USER_SPHERE, 0.0, 0
        self.seeds = None
        self.reset()

    def reset(self):
        self.sel, self.n = 0, 0
        self.gl.L.glref_clear_fbo(self.accfb[0], 0, 0, 0, 1)
        self.seed_idx = 0

    def _common(self, prog):
        gl = self.gl
        sun = O.sun_direction(self.sun_az, self.sun_alt)
        gl.bind(0, self.rdir); gl.bind(1, self.rstart); gl.bind(2, self.tbo[0], buffer_tex=True)
        gl.set_uniforms(prog, RDir=0, RStart=1, BVH=2, SunDirAlt=[sun[0], sun[1], sun[2], self.sun_alt],
                        SunDirectLightingEnabled=int(self.sun_on), UserSphere=list(self.us),
                        UserSphereFlags=("u", self.us_flags))

    def cam_rays(self):
        return self.gl.read(self.rstart, self.W, self.H), self.gl.read(self.rdir, self.W, self.H)

    def direct(self):
        self._common(self.p.direct)
        self.gl.draw(self.p.direct, self.outfb, self.W, self.H)
        self.gl.finish()
        return self.gl.read(self.out, self.W, self.H)

    def pt_pass(self, npaths, rand_seed):
        gl = self.gl
        src, dst = self.sel, self.sel ^ 1
        self._common(self.p.pt)
        gl.bind(3, self.acc[src])
        gl.set_uniforms(self.p.pt, PrevRadiance=3, NumPathsPerPixel=int(npaths), PixelSize=float(self.cam[12]),
                        CameraPos=self.cam[0:3], UserSphereEm=[self.us_em] * 3, RandSeed=[float(x) for x in rand_seed])
        gl.draw(self.p.pt, self.accfb[dst], self.W, self.H)
        gl.finish()
        self.n += npaths
        self.sel ^= 1
        self.last = dst
        return gl.read(self.acc[dst], self.W, self.H)


def scene_tree(name):
    prims = {"box": S.box_scene, "scene_p": S.scene_p, "scene_d": S.scene_d, "cluster": S.cluster_scene, "tree": S.tree_scene,
             "dragon871k": lambda: S.scene_d(660, 660),
             "lattice": S.lattice_scene, "lattice_big": lambda: S.lattice_scene(seed=5, n=3000),
             "scene_pc": lambda: S.scene_p(seed=3, nspheres=96, ndiscs=24, ncones=64)}[name]()
    tree, depth = O.build_bvh(prims)
    return prims, tree, depth


def default_cam(which=S.DEFAULT_CAMERA):
    c = dict(which)
    c["dir"] = S.camera_dir(c)
    return c


def gen_camrays(gl):
    progs = RefPrograms(gl)
    _, tree, _ = scene_tree("box")
    for W, H in [(64, 36), (37, 23)]:
        r = RefRenderer(gl, progs, W, H, default_cam(), tree)
        rs, rd = r.cam_rays()
        save("camrays_%dx%d" % (W, H), cam=r.cam, rstart=rs[..., :3].copy(), rdir=rd[..., :3].copy())


def gen_traverse(gl):
    progs = RefPrograms(gl)
    rng = np.random.RandomState(21)
    for name, (W, H), camsel in [("box", (96, 96), S.DEFAULT_CAMERA), ("scene_pc", (96, 54), S.DEFAULT_CAMERA),
                                 ("scene_d", (128, 72), S.BENCH_CAMERA)]:
        _, tree, _ = scene_tree(name)
        r = RefRenderer(gl, progs, W, H, default_cam(camsel), tree)
        rs, rd = r.cam_rays()
        rs, rd = rs.reshape(-1, 4), rd.reshape(-1, 4)
        body = ("float pos; vec3 p, n; int t; bool ush; CheckIntersectionInclUserSphere(i0.xyz, i1.xyz, BVH, vec4(-0.4, 0, 0.2, 0), pos, p, n, t, ush);"
                "if (t >= 0) { O0 = vec4(pos, p); O1 = vec4(n, float(t) + (ush ? 0.5 : 0.0)); } else { O0 = vec4(-1, 0, 0, 0); O1 = vec4(0, 0, 0, -1); }")
        objs = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl",
                "bvh_intersection.glsl", "intersection.glsl"]
        o0, o1 = glref.run_probe_big(gl, body, objs, [rs, rd], 2, bvh=tree, decls=D_INCL, chunk=4096)
        # secondary rays: from the primary hit points, random directions (incoherent) + sun direction
        hit = o1[:, 3] >= 0
        n2 = min(4096, int(hit.sum()))
        idx = np.flatnonzero(hit)[rng.permutation(int(hit.sum()))[:n2]]
        rs2 = np.zeros((n2, 4), np.float32); rs2[:, :3] = o0[idx, 1:4]
        rd2 = np.zeros((n2, 4), np.float32)
        d = unit(rng.normal(size=(n2, 3))); d *= np.sign((d * o1[idx, :3]).sum(1, keepdims=True)); rd2[:, :3] = d
        sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
        rd2[n2 // 2:, :3] = sun
        s0, s1 = glref.run_probe_big(gl, body, objs, [rs2, rd2], 2, bvh=tree, decls=D_INCL, chunk=4096)
        save("traverse_" + name, scene=name, W=W, H=H, cam=r.cam, rs=rs[:, :3].copy(), rd=rd[:, :3].copy(), o0=o0, o1=o1,
             rs2=rs2[:, :3].copy(), rd2=rd2[:, :3].copy(), s0=s0, s1=s1)


def gen_traverse_wild(gl):
    """CheckIntersectionInclUserSphere with hostile rays on a regular tree (scene_pc) and with regular + hostile rays on the trees
    of wild scenes (irregular boxes, degenerate 200-level chains: random_wild_case 42874 and 7; random_wild2_case 5), user sphere
    of radius 0.25 included. The tree travels in the fixture."""
    rng = np.random.RandomState(9201)
    # (the user sphere travels as an input, not as a literal: a literal lets the compiler fold `- 0` and `0.25 * 0.25` into
    #  SphereIntersection, which the renderer's uniform never allows)
    body = ("float pos; vec3 p, n; int t; bool ush; CheckIntersectionInclUserSphere(i0.xyz, i1.xyz, BVH, i2, pos, p, n, t, ush);"
            "if (t >= 0) { O0 = vec4(pos, p); O1 = vec4(n, float(t) + (ush ? 0.5 : 0.0)); } else { O0 = vec4(-1, 0, 0, 0); O1 = vec4(0, 0, 0, -1); }")
    objs = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl", "bvh_intersection.glsl", "intersection.glsl"]
    trees = [("scene_pc", scene_tree("scene_pc")[1]), ("wild_42874", O.build_bvh(S.random_wild_case(42874)["prims"])[0]),
             ("wild_7", O.build_bvh(S.random_wild_case(7)["prims"])[0]), ("wild2_5", O.build_bvh(S.random_wild2_case(5)["prims"])[0])]
    for name, tree in trees:
        n = 4096
        rs = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32); rs[:, 2] = np.abs(rs[:, 2])
        aim = rng.uniform(-1.2, 1.2, (n, 3)); aim[:, 2] = np.abs(aim[:, 2])
        rd = ((aim - rs) * rng.uniform(0.05, 2.0, (n, 1))).astype(np.float32)
        inject(rng, [rs[n // 2:], rd[n // 2:]], 0.7)   # second half: hostile rays
        us = np.tile(np.array([-0.4, 0, 0.2, 0.25], np.float32), (n, 1))
        o0, o1 = glref.run_probe_big(gl, body, objs, [pad4(rs), pad4(rd), us], 2, bvh=tree, decls=D_INCL, chunk=4096)
        print("traverse_wild %s: %d of %d rays hit" % (name, int((o1[:, 3] >= 0).sum()), n))
        save("traverse_wild_" + name, tree=tree, rs=rs, rd=rd, o0=o0, o1=o1)


def gen_order_rays(gl):
    """The rays of tests/golden/order_rays.npz (found on the GPU by tools/order_rays.py: closest-hit queries on which a walk in
    another order than the reference's returned another primitive, because a box on the way reports an entry parameter beyond the
    hits inside it) through the reference's CheckIntersectionInclUserSphere on llvmpipe: what the reference itself answers."""
    body = ("float pos; vec3 p, n; int t; bool ush; CheckIntersectionInclUserSphere(i0.xyz, i1.xyz, BVH, i2, pos, p, n, t, ush);"
            "if (t >= 0) { O0 = vec4(pos, p); O1 = vec4(n, float(t) + (ush ? 0.5 : 0.0)); } else { O0 = vec4(-1, 0, 0, 0); O1 = vec4(0, 0, 0, -1); }")
    objs = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl", "bvh_intersection.glsl", "intersection.glsl"]
    g = dict(np.load(os.path.join(OUT, "order_rays.npz")))
    for name, scene in (("cfg3", "scene_d"), ("tree", "tree")):
        _, tree, _ = scene_tree(scene)
        rs, rd = g[name + "_rs"], g[name + "_rd"]
        us = np.zeros((len(rs), 4), np.float32)
        o0, o1 = glref.run_probe_big(gl, body, objs, [pad4(rs), pad4(rd), us], 2, bvh=tree, decls=D_INCL, chunk=4096)
        g[name + "_o0"], g[name + "_o1"] = o0, o1
        print("order_rays %s:" % name, o0[:, 0], o1[:, 3])
    save("order_rays", **g)


def gen_traverse_leaves(gl):
    """CheckIntersectionInclUserSphere on trees with leaves of other sizes than the default build makes (minPrimitivesPerNode 5;
    depth limits 4, 6 and 1 = the whole scene in the root leaf): the device walks such leaves with its counting loop instead of
    the fetch-at-once paths. Regular rays + hostile ones; the tree travels in the fixture (same format as traverse_wild)."""
    rng = np.random.RandomState(9311)
    body = ("float pos; vec3 p, n; int t; bool ush; CheckIntersectionInclUserSphere(i0.xyz, i1.xyz, BVH, i2, pos, p, n, t, ush);"
            "if (t >= 0) { O0 = vec4(pos, p); O1 = vec4(n, float(t) + (ush ? 0.5 : 0.0)); } else { O0 = vec4(-1, 0, 0, 0); O1 = vec4(0, 0, 0, -1); }")
    objs = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl", "bvh_intersection.glsl", "intersection.glsl"]
    soup = [(S.DISC, [0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 30.0])]
    for _ in range(200):
        o = rng.uniform(-1.5, 1.5, 3); o[2] = abs(o[2]) * 0.5 + 0.05
        soup.append((S.TRIANGLE, [float(np.float32(v)) for v in np.concatenate([o, o + rng.uniform(-0.5, 0.5, 3), o + rng.uniform(-0.5, 0.5, 3)])]))
    trees = [("pc_min5", O.build_bvh(scene_tree("scene_pc")[0], min_prims=5)[0]), ("pc_levels4", O.build_bvh(scene_tree("scene_pc")[0], max_levels=4)[0]),
             ("p_root_leaf", O.build_bvh(S.scene_p(), max_levels=1)[0]), ("soup_levels6", O.build_bvh(soup, max_levels=6)[0])]
    for name, tree in trees:
        n = 2048
        rs = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32); rs[:, 2] = np.abs(rs[:, 2])
        aim = rng.uniform(-1.2, 1.2, (n, 3)); aim[:, 2] = np.abs(aim[:, 2])
        rd = ((aim - rs) * rng.uniform(0.05, 2.0, (n, 1))).astype(np.float32)
        inject(rng, [rs[3 * n // 4:], rd[3 * n // 4:]], 0.7)   # last quarter: hostile rays
        us = np.tile(np.array([-0.4, 0, 0.2, 0.25], np.float32), (n, 1))
        o0, o1 = glref.run_probe_big(gl, body, objs, [pad4(rs), pad4(rd), us], 2, bvh=tree, decls=D_INCL, chunk=2048)
        print("traverse_leaves %s: %d quads, %d of %d rays hit" % (name, len(tree), int((o1[:, 3] >= 0).sum()), n))
        save("traverse_wild_" + name, tree=tree, rs=rs, rd=rd, o0=o0, o1=o1)


def gen_frames(gl):
    seeds = O.randseeds(16)
    for name, (W, H), camsel, segs in [("box", (128, 128), S.DEFAULT_CAMERA, [5, 1, 4, 8]),
                                       ("scene_pc", (128, 72), S.DEFAULT_CAMERA, [5]),
                                       ("scene_d", (128, 72), S.BENCH_CAMERA, [5, 8])]:
        _, tree, _ = scene_tree(name)
        for ms in segs:
            progs = RefPrograms(gl, ms)
            r = RefRenderer(gl, progs, W, H, default_cam(camsel), tree)
            out = {}
            if ms == 5:
                out["direct"] = r.direct()[..., :3].copy()
                r.sun_on = 0
                out["direct_nosun"] = r.direct()[..., :3].copy()
                r.sun_on = 1
            r.reset()
            npass = 8 if name == "box" and ms == 5 else 2
            for k in range(npass):
                acc = r.pt_pass(1, seeds[k])
                if k == 0:
                    out["pt_pass1"] = acc[..., :3].copy()
            out["pt_acc"] = acc[..., :3].copy()
            # a 3-paths-per-pass run (NumPathsPerPixel loop, path_tracing.glsl:154)
            if ms == 5:
                r.reset()
                out["pt_3paths"] = r.pt_pass(3, seeds[0])[..., :3].copy()
            save("frames_%s_seg%d" % (name, ms), scene=name, W=W, H=H, cam=r.cam, max_segments=ms, npasses=npass,
                 seeds=seeds, **out)
    # user sphere variants on the Box scene (emissive / specular / fuzzy), r = 0.25
    _, tree, _ = scene_tree("box")
    progs = RefPrograms(gl)
    for tag, flags, em in [("em", 1, 2.0), ("spec", 2, 0.0), ("fuzzy", 6, 0.0), ("diffuse", 0, 0.0)]:
        r = RefRenderer(gl, progs, 96, 96, default_cam(), tree)
        r.us, r.us_em, r.us_flags = (-0.4, 0.0, 0.25, 0.25), em, flags
        out = {"direct": r.direct()[..., :3].copy()}
        r.reset()
        out["pt_pass1"] = r.pt_pass(1, seeds[0])[..., :3].copy()
        out["pt_acc"] = r.pt_pass(1, seeds[1])[..., :3].copy()
        save("frames_box_usph_" + tag, scene="box", W=96, H=96, cam=r.cam, user_sphere=np.array(r.us, np.float32),
             user_sphere_em=em, user_sphere_flags=flags, seeds=seeds, **out)


def gen_fuzz(gl):
    """Random cases of gpuart_amd.synth_scenes.random_case (degenerate primitives, duplicates, random cameras / user
    sphere / Sun / depth) rendered by the reference's shaders: direct lighting + `passes` accumulated path-tracing passes.
    Pins the oracle on inputs nobody hand-picked."""
    progs = {}
    out = {}
    cases = list(range(24))
    for seed in cases:
        case = S.random_case(seed)
        tree, _ = O.build_bvh(case["prims"])
        ms = case["max_segments"]
        if ms not in progs:
            progs[ms] = RefPrograms(gl, ms)
        r = RefRenderer(gl, progs[ms], case["W"], case["H"], case["cam"], tree)
        r.us, r.us_em, r.us_flags = case["user_sphere"], case["us_em"], case["us_flags"]
        r.sun_az, r.sun_alt, r.sun_on = case["sun_az"], case["sun_alt"], int(case["sun_on"])
        seeds = O.randseeds(case["passes"], seed=5489 + seed)
        out["direct_%d" % seed] = r.direct()[..., :3].copy()
        r.reset()
        for k in range(case["passes"]):
            acc = r.pt_pass(case["npaths"], seeds[k])
        out["pt_acc_%d" % seed] = acc[..., :3].copy()
        print("  fuzz case %d: %d primitives, %dx%d, depth %d, flags %d" % (seed, len(case["prims"]), case["W"], case["H"], ms, case["us_flags"]))
    save("fuzz_frames", cases=np.array(cases, np.int32), **out)


def row_checksums(img):
    """Per row and channel: sum of the float32 bit patterns (uint64) — order-independent inside a row, exact."""
    return img[..., :3].view(np.uint32).astype(np.uint64).sum(axis=1)


def gen_fullsize(gl):
    """BASELINE cfg3 / cfg4 at their full sizes (Scene D, depth 8, benchmark camera; 1920x1080 and 3840x2160) rendered by
    the reference's shaders; stored as per-row checksums of the bit patterns (a 1080p RGBA32F frame is 33 MB)."""
    _, tree, _ = scene_tree("scene_d")
    progs = RefPrograms(gl, 8)
    for tag, W, H, npass in (("1080p", 1920, 1080, 2), ("4k", 3840, 2160, 1)):
        r = RefRenderer(gl, progs, W, H, default_cam(S.BENCH_CAMERA), tree)
        seeds = O.randseeds(2)
        out = {"direct": row_checksums(r.direct())}
        r.reset()
        for k in range(npass):
            acc = r.pt_pass(1, seeds[k])
            out["pt_acc%d" % (k + 1)] = row_checksums(acc)
        save("fullsize_scene_d_" + tag, W=W, H=H, cam=r.cam, max_segments=8, seeds=seeds, npasses=npass, **out)
        for t in (r.rstart, r.rdir, r.out, *r.acc):
            gl.L.glref_delete_tex(t)


def gen_scene_p(gl):
    """BASELINE cfg1 and cfg2 on Scene P (floor disc + 256 spheres + 16 tilted discs), default camera:
    cfg1 = 256x256 direct lighting, the whole frame as bit patterns (+ two depth-4 path-tracing passes at that size);
    cfg2 = 1920x1080 path tracing depth 4, 1 path/pixel/pass: direct lighting + two passes as per-row checksums."""
    _, tree, _ = scene_tree("scene_p")
    seeds = O.randseeds(16)
    progs = RefPrograms(gl, 4)
    r = RefRenderer(gl, progs, 256, 256, default_cam(), tree)
    out = {"direct": r.direct()[..., :3].copy()}
    r.reset()
    out["pt_pass1"] = r.pt_pass(1, seeds[0])[..., :3].copy()
    out["pt_acc"] = r.pt_pass(1, seeds[1])[..., :3].copy()
    save("frames_scene_p_seg4", scene="scene_p", W=256, H=256, cam=r.cam, max_segments=4, npasses=2, seeds=seeds, **out)
    r = RefRenderer(gl, progs, 1920, 1080, default_cam(), tree)
    out = {"direct": row_checksums(r.direct())}
    r.reset()
    for k in range(2):
        out["pt_acc%d" % (k + 1)] = row_checksums(r.pt_pass(1, seeds[k]))
    save("fullsize_scene_p_1080p", W=1920, H=1080, cam=r.cam, max_segments=4, seeds=seeds[:2], npasses=2, **out)


def gen_cluster_tree(gl):
    """The reference's two primitive-list scenes (InitCluster: 100k spheres, InitTree: cones + spheres; src/scenes.cpp:69-103)
    on seeded synthetic stand-ins of the absent data files: direct lighting + two path-tracing passes, 128x72, both cameras."""
    seeds = O.randseeds(16)
    progs = RefPrograms(gl)
    for name in ("cluster", "tree"):
        _, tree, depth = scene_tree(name)
        for tag, camsel in (("", S.DEFAULT_CAMERA), ("_near", S.NEAR_CAMERAS[name])):
            r = RefRenderer(gl, progs, 128, 72, default_cam(camsel), tree)
            out = {"direct": r.direct()[..., :3].copy()}
            r.reset()
            out["pt_pass1"] = r.pt_pass(1, seeds[0])[..., :3].copy()
            out["pt_acc"] = r.pt_pass(1, seeds[1])[..., :3].copy()
            save("frames_%s%s_seg5" % (name, tag), scene=name, W=128, H=72, cam=r.cam, max_segments=5, npasses=2, seeds=seeds,
                 bvh_depth=depth, **out)


def gen_lattice(gl):
    """Scenes of coplanar, overlapping axis-aligned triangles and discs and coincident spheres (synth_scenes.lattice_scene): which of
    two coincident surfaces a ray reports hinges on the order in which the reference's walk meets them. 401 and 3 001 primitives,
    direct lighting + two path-tracing passes (+ one pass of 3 paths for the small one), 128x72, default camera.
    The large scene's tree is 402 levels deep (its primitives' box centres tie all the time) and a ray walks thousands of nodes:
    with 3 paths per pass a fragment exceeds 65 535 loop iterations, where llvmpipe (gallivm's loop limiter: one budget for all loops
    of a shader invocation) breaks the reference's traversal loop off — the frame is then llvmpipe's, not the reference's (measured:
    0 / 101 / 2 361 of 9 216 pixels differ from the oracle with 1 / 2 / 3 paths per pass, oracle/README.md). Fixtures stay below."""
    seeds = O.randseeds(16)
    progs = RefPrograms(gl)
    for name in ("lattice", "lattice_big"):
        _, tree, depth = scene_tree(name)
        r = RefRenderer(gl, progs, 128, 72, default_cam(S.DEFAULT_CAMERA), tree)
        out = {"direct": r.direct()[..., :3].copy()}
        r.reset()
        out["pt_pass1"] = r.pt_pass(1, seeds[0])[..., :3].copy()
        out["pt_acc"] = r.pt_pass(1, seeds[1])[..., :3].copy()
        if name == "lattice":
            r.reset()
            out["pt_3paths"] = r.pt_pass(3, seeds[0])[..., :3].copy()
        save("frames_%s_seg5" % name, scene=name, W=128, H=72, cam=r.cam, max_segments=5, npasses=2, seeds=seeds, bvh_depth=depth, **out)


def gen_dragon871k(gl):
    """The reference's largest scene as a stand-in of its size (main.cpp:321 "dragon 871k": 871 200 triangles; here the displaced torus
    at 660 x 660 quads + floor disc; 75 MB of tree on the device, which no longer fits the L2s): benchmark camera, depth 8,
    direct lighting + two passes at 128x72 through the reference's shaders."""
    seeds = O.randseeds(16)
    _, tree, depth = scene_tree("dragon871k")
    progs = RefPrograms(gl, 8)
    r = RefRenderer(gl, progs, 128, 72, default_cam(S.BENCH_CAMERA), tree)
    out = {"direct": r.direct()[..., :3].copy()}
    r.reset()
    out["pt_pass1"] = r.pt_pass(1, seeds[0])[..., :3].copy()
    out["pt_acc"] = r.pt_pass(1, seeds[1])[..., :3].copy()
    save("frames_dragon871k_seg8", scene="dragon871k", W=128, H=72, cam=r.cam, max_segments=8, npasses=2, seeds=seeds, bvh_depth=depth, **out)


# ---- the adversary of the nearest-child-first walk (round 5) ----------------------------------------------------------------
# A walk that visits a node's children in another order than the reference's (lower child first, shaders/bvh_intersection.glsl:432-441)
# and prunes on the closest hit so far returns the reference's winner only if every primitive the reference TESTS and this walk does
# not has a parameter beyond the winner's. The reference accepts whatever parameter its intersector computes — and Moeller-Trumbore
# (shaders/triangle.glsl:50-76) at a grazing angle below ~1e-5 rad computes det and the numerator from sums that cancel to a few
# ulps: t = (a few ulps) / (a few ulps), any small dyadic number, while u and v (as arbitrary) happen to pass. Such a hit can lie far
# IN FRONT of its own leaf box. Put a surface S between that phantom hit and the box: the reference, arriving at the triangle's
# leaf first with nothing closer, tests it and keeps the phantom; a walk that sees S first prunes the leaf (entered beyond S by more
# than any band) and never tests the triangle — no certificate computed from what the walk DID test can notice.
def _nf_parse(tree):
    q = tree.view(np.uint32).reshape(-1, 4); f = tree.reshape(-1, 4)
    nodes = {}

    def rd(a):
        flags = int(q[a + 2, 0])
        n = {"min": f[a, :3].copy(), "max": f[a + 1, :3].copy(), "leaf": bool(flags >> 31 & 1)}
        if n["leaf"]:
            prims, p = [], a + 3
            for _ in range(flags & 0x1fffffff):
                t = int(q[p, 0]); nd = t + 1
                prims.append((p, t, f[p + 1:p + 1 + nd].copy())); p += 1 + nd
            n["prims"] = prims
        else:
            n["lo"], n["hi"] = int(q[a + 2, 1]), int(q[a + 2, 2]); rd(n["lo"]); rd(n["hi"])
        nodes[a] = n
    rd(0)
    return nodes


def _nf_walk(nodes, o, d, band=np.float32(1.00390625)):
    """Model of the device's nearest-child-first closest-hit walk WITH its certificate (gpuart_amd/csrc/hip/device_scene.h: trav_step_box,
    take_hit, trav_settle), on the oracle's box test and intersectors: (closest, winner's address, whether the certificate asks for a second walk)."""
    f32 = np.float32
    v4 = lambda v: np.append(v, 0)[None].astype(f32)

    def box(n):
        a = O.aabb(v4(o), v4(d), v4(n["min"]), v4(n["max"]))[0]
        rdiv = (f32(1) / d).astype(f32)
        k0, k1 = ((n["min"] - o).astype(f32) * rdiv).astype(f32), ((n["max"] - o).astype(f32) * rdiv).astype(f32)
        slab = np.max(np.minimum(k0, k1))
        hit = a[0] > 0
        return hit, a[1], bool(hit and a[1] != -1 and a[1] != min(slab, f32(1e19)))

    def prim(pr):
        _, t, data = pr
        if t == 2: r = O.triangle(v4(o), v4(d), data[0][None], data[1][None], data[2][None])[0][0, 0]
        elif t == 1: r = O.disc(v4(o), v4(d), data[0][None], data[1][None])[0][0, 0]
        else: r = O.sphere(v4(o), v4(d), data[0][None])[0][0, 0]
        return r if r >= f32(1e-4) else f32(-1)
    closest = second = f32(1e19); win, flags, odd = None, 0, False
    h, e, od = box(nodes[0])
    if not h:
        return closest, None, False
    odd |= od
    stack, cur = [], (0, e)
    while True:
        a, entry = cur; n = nodes[a]; nxt = None
        if n["leaf"]:
            for pr in n["prims"]:
                t = prim(pr)
                if t > 0:
                    w = t < closest or (t == closest and (win is None or pr[0] < win))
                    second = min(second, closest if w else t)
                    if w:
                        closest, win, flags = t, pr[0], (1 if entry > t else 0) | (2 if entry > t * band else 0)
        else:
            hl, el, ol = box(nodes[n["lo"]]); hh, eh, oh = box(nodes[n["hi"]]); odd |= ol | oh
            en, ef, rn, rf = (el if hl else f32(3e38)), (eh if hh else f32(3e38)), n["lo"], n["hi"]
            if ef < en: en, ef, rn, rf = ef, en, rf, rn
            if not ef > closest * band: stack.append((rf, entry, ef))
            if not en > closest * band: nxt = (rn, en)
        while nxt is None:
            if not stack:
                return closest, win, bool(odd or (win is not None and ((flags & 2) or ((flags & 1) and second <= closest * band))))
            rf, pe, he = stack.pop()
            if pe > closest * band or he > closest * band: continue
            nxt = (rf, he)
        cur = nxt


def gen_order_adversary(gl):
    """tests/golden/order_adversary.npz: scenes of four primitives + one ray each on which the reference (its own GLSL, here) returns a
    triangle's PHANTOM hit — a parameter far in front of the triangle's own box, from cancelled sums at a grazing angle — while a
    nearest-child-first walk, certificate included, returns the surface that stands between (model above; the device's walk is held
    to it in tests/test_gpu_parity.py). Search: seeded; the oracle's intersector finds candidate rays, the reference's GLSL has the say."""
    f32 = np.float32
    rng = np.random.RandomState(20261005)

    def rot():
        q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    body = ("float pos; vec3 p, n; int t; bool ush; CheckIntersectionInclUserSphere(i0.xyz, i1.xyz, BVH, i2, pos, p, n, t, ush);"
            "if (t >= 0) { O0 = vec4(pos, p); O1 = vec4(n, float(t) + (ush ? 0.5 : 0.0)); } else { O0 = vec4(-1, 0, 0, 0); O1 = vec4(0, 0, 0, -1); }")
    objs = ["sphere.glsl", "disc.glsl", "triangle.glsl", "cone.glsl", "common.glsl", "noise.glsl", "bvh_intersection.glsl", "intersection.glsl"]
    cases = {}
    while len(cases) < 16 * 6:
        eps, s, R, c = 10 ** rng.uniform(-6.5, -5), 10 ** rng.uniform(-1.5, -0.5), rot(), rng.uniform(-1, 1, 3)
        V = (np.array([[0, 0, 0], [s, 0.1 * s, 0], [0.3 * s, 0.9 * s, 0]]) @ R.T + c).astype(f32)   # a well-shaped triangle, any orientation
        n = 100000
        ol = np.zeros((n, 3)); ol[:, 0] = -2.0 + rng.uniform(-0.5, 0.5, n); ol[:, 1] = rng.uniform(0, s, n); ol[:, 2] = rng.uniform(-4, 4, n) * eps
        tgt = np.zeros((n, 3)); tgt[:, 0] = rng.uniform(0, s, n); tgt[:, 1] = rng.uniform(0, s, n)
        dl = tgt - ol; dl[:, 2] += rng.uniform(-1, 1, n) * eps * 0.5                                     # rays that graze its plane by ~eps rad
        o, d = pad4((ol @ R.T + c).astype(f32)), pad4((dl @ R.T).astype(f32))
        t = O.triangle(o, d, *[np.tile(pad4(V[k][None]), (n, 1)) for k in range(3)])[0][:, 0]
        a = O.aabb(o, d, np.tile(pad4(V.min(0)[None]), (n, 1)), np.tile(pad4(V.max(0)[None]), (n, 1)))
        bad = np.flatnonzero((t > 0) & (a[:, 0] > 0) & (a[:, 1] > 0) & (t * 1.05 < a[:, 1]))            # the hit lies 5 % and more in front of the triangle's box
        if not len(bad):
            continue
        k = bad[np.argmax(a[bad, 1] / t[bad])]
        ray_o, ray_d, tp, eT = o[k, :3], d[k, :3], t[k], a[k, 1]
        for _ in range(40):
            tS = f32(rng.uniform(tp * 1.002, eT / 1.01))                                                   # S: a disc across the ray between phantom and box
            cS = ray_o.astype(np.float64) + tS * ray_d.astype(np.float64)
            tri = lambda cc, sz: (2, list((cc + rng.normal(size=3) * sz)) + list((cc + rng.normal(size=3) * sz)) + list((cc + rng.normal(size=3) * sz)))
            prims = [(2, list(V.reshape(-1).astype(float))), tri(V.mean(0) + rng.normal(size=3) * 0.3, 0.01),
                     (1, list(cS) + list(ray_d / np.linalg.norm(ray_d)) + [float(rng.uniform(0.005, 0.05))]), tri(cS + rng.normal(size=3) * 0.3, 0.01)]
            tree, _ = O.build_bvh(prims)
            r0, r1 = O.traverse(tree, pad4(ray_o[None]), pad4(ray_d[None]), (0, 0, 0, 0))
            cl, win, again = _nf_walk(_nf_parse(tree), ray_o, ray_d)
            if r1[0, 3] >= 0 and win is not None and r0[0, 0] != cl and not again:
                g0, g1 = glref.run_probe_big(gl, body, objs, [pad4(ray_o[None]), pad4(ray_d[None]), np.zeros((1, 4), f32)], 2, bvh=tree, decls=D_INCL, chunk=4096)
                if g0[0, 0] != cl:   # the reference's GLSL agrees that the walk's answer is not the reference's
                    i = len(cases) // 6
                    cases.update({"tree%d" % i: tree, "rs%d" % i: ray_o, "rd%d" % i: ray_d, "o0_%d" % i: g0[0], "o1_%d" % i: g1[0], "nf%d" % i: f32(cl)})
                    print("order_adversary %2d: reference t = %.6f (type %g), nearest-first walk t = %.6f; the triangle's box is entered at %.6f" % (i, g0[0, 0], g1[0, 3], cl, eT))
                break
    save("order_adversary", n=len(cases) // 6, **cases)


def gen_order_adversary_frames(gl):
    """tests/golden/order_adversary_frame_*.npz: the phantom hit of gen_order_adversary END TO END, through the reference's path-tracing
    program. A small frame with the default camera; the first-segment ray of one pixel (the jittered camera ray, path_tracing.glsl:141-175,
    from the oracle's restatement) is the ray a triangle is laid under at a grazing angle until the reference's intersector returns a
    phantom; a disc between phantom and triangle, two fillers. The reference's own programs render the frame: the pixel shows the TRIANGLE
    (its phantom), a walk that prunes in another order shows the disc. Expected values: the reference's GLSL (pt pass 1 + accumulation of 2)."""
    f32 = np.float32
    rng = np.random.RandomState(20261006)
    progs = RefPrograms(gl)
    W, H = 24, 16
    camd = default_cam()
    cam = O.camera(camd["pos"], camd["dir"], camd["up"], camd["fov_y"], camd["screen_dist"], W, H)
    sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
    P = O.make_params(sun, S.SUN_ALTITUDE, True, S.USER_SPHERE, 0.0, 0, float(cam[12]), cam[0:3], 5, 0.01)
    seeds = O.randseeds(2)
    rs_all, rd_all = O.first_segment_rays(cam, W, H, P, seeds[0], 0)
    made = 0
    for attempt in range(4000):
        if made >= 3:
            break
        px, py = int(rng.randint(4, W - 4)), int(rng.randint(4, H - 4))
        o, d = rs_all[py, px, :3].copy(), rd_all[py, px, :3].copy()
        u = d.astype(np.float64) / np.linalg.norm(d)
        a = np.cross(u, rng.normal(size=3)); a /= np.linalg.norm(a)
        b = np.cross(u, a)
        n = 60000
        eps = 10 ** rng.uniform(-6.5, -5, n); s_ = 10 ** rng.uniform(-1.3, -0.5, n); Ld = rng.uniform(1.0, 3.0, n); a0 = rng.uniform(0.25, 0.55, n)
        phi = rng.uniform(0, 2 * np.pi, n)
        v = np.cos(phi)[:, None] * a + np.sin(phi)[:, None] * b
        w = -np.sin(phi)[:, None] * a + np.cos(phi)[:, None] * b
        loc = np.array([[0.0, -0.4], [1.0, -0.1], [0.3, 0.6]])
        V = np.zeros((n, 3, 3))
        for k in range(3):
            al, be = loc[k, 0] * s_, loc[k, 1] * s_
            hgt = eps * (al - a0 * s_)
            V[:, k, :] = o.astype(np.float64) + u * (Ld + al)[:, None] + v * be[:, None] + w * hgt[:, None]
        V = V.astype(f32)
        O4, D4 = np.tile(pad4(o[None]), (n, 1)), np.tile(pad4(d[None]), (n, 1))
        t = O.triangle(O4, D4, pad4(V[:, 0]), pad4(V[:, 1]), pad4(V[:, 2]))[0][:, 0]
        bx = O.aabb(O4, D4, pad4(V.min(1)), pad4(V.max(1)))
        bad = np.flatnonzero((t > 0) & (bx[:, 0] > 0) & (bx[:, 1] > 0) & (t * 1.05 < bx[:, 1]))
        if not len(bad):
            continue
        k = bad[np.argmax(bx[bad, 1] / t[bad])]
        Vt, tp, eT = V[k], t[k], bx[k, 1]
        for _ in range(30):
            tS = f32(rng.uniform(tp * 1.002, eT / 1.01))
            cS = o.astype(np.float64) + tS * d.astype(np.float64)
            tri = lambda cc, sz: (2, list((cc + rng.normal(size=3) * sz)) + list((cc + rng.normal(size=3) * sz)) + list((cc + rng.normal(size=3) * sz)))
            prims = [(2, list(Vt.reshape(-1).astype(float))), tri(Vt.mean(0) + rng.normal(size=3) * 0.3, 0.01),
                     (1, list(cS) + list(u) + [float(rng.uniform(0.01, 0.04))]), tri(cS + rng.normal(size=3) * 0.3, 0.01)]
            tree, _ = O.build_bvh(prims)
            r0, r1 = O.traverse(tree, pad4(o[None]), pad4(d[None]), (0, 0, 0, 0))
            cl, win, again = _nf_walk(_nf_parse(tree), o, d)
            if not (r1[0, 3] == 2.0 and win is not None and r0[0, 0] != cl and not again):
                continue
            r = RefRenderer(gl, progs, W, H, camd, tree)
            r.us = (S.USER_SPHERE[0], S.USER_SPHERE[1], S.USER_SPHERE[2], 0.0)
            p1 = r.pt_pass(1, seeds[0])[..., :3].copy()
            acc = r.pt_pass(1, seeds[1])[..., :3].copy()
            # the oracle's frame must be the reference's, and the pixel must be the triangle's: re-render with the triangle removed — the disc shows
            acc_o = np.zeros((H, W, 4), f32)
            O.pt_pass(tree, cam, W, H, P, seeds[0], 1, acc_o)
            if not (acc_o[..., :3].view(np.uint32) == p1.view(np.uint32)).all():
                print("order_adversary_frames: oracle != reference on a candidate (kept out; investigate)"); continue
            tree2, _ = O.build_bvh(prims[1:])
            acc2 = np.zeros((H, W, 4), f32)
            O.pt_pass(tree2, cam, W, H, P, seeds[0], 1, acc2)
            if (acc2[py, px, :3].view(np.uint32) == p1[py, px].view(np.uint32)).all():
                continue  # (the pixel looks the same with and without the triangle: not a visible phantom)
            save("order_adversary_frame_%d" % made, tree=tree, W=W, H=H, cam=cam, seeds=seeds, max_segments=5, pixel=np.array([px, py]),
                 pt_pass1=p1, pt_acc=acc, nearest_first_t=f32(cl), reference_t=r0[0, 0])
            print("order_adversary_frame_%d: pixel (%d, %d): the reference's first query returns the triangle's phantom at t = %.6f, a nearest-first walk the disc at %.6f"
                  % (made, px, py, r0[0, 0], cl))
            made += 1
            break
    assert made == 3, made


SECTIONS = dict(dragon871k=gen_dragon871k, lattice=gen_lattice, scene_p=gen_scene_p, cluster_tree=gen_cluster_tree, hash=gen_hash, llvmpipe_math=gen_llvmpipe_math, hemisphere=gen_hemisphere, sphere=gen_sphere,
                disc=gen_disc, triangle=gen_triangle, cone=gen_cone, aabb=gen_aabb, aabb_irregular=gen_aabb_irregular, intersect_wild=gen_intersect_wild, shade_wild=gen_shade_wild, traverse_wild=gen_traverse_wild, traverse_leaves=gen_traverse_leaves, order_rays=gen_order_rays, order_adversary=gen_order_adversary, order_adversary_frames=gen_order_adversary_frames, sky=gen_sky, uv=gen_uv, camrays=gen_camrays,
                traverse=gen_traverse, frames=gen_frames, fuzz=gen_fuzz, fullsize=gen_fullsize)

if __name__ == "__main__":
    if not glref.available():
        sys.exit("needs /root/reference/shaders and oracle/_ref/libglref.so (make -C oracle)")
    gl = glref.GLRef()
    print("GL_RENDERER =", gl.renderer())
    for s in (sys.argv[1:] or list(SECTIONS)):
        SECTIONS[s](gl)
