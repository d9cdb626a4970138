"""ctypes front-end of oracle/liboracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg — never by the product package (gpuart_amd/), which must fail loudly when
its HIP library is missing instead of falling back to this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")

SPHERE, DISC, TRIANGLE, CONE = 0, 1, 2, 3


class PrimDesc(C.Structure):
    _fields_ = [("type", C.c_int), ("f", C.c_float * 9)]


class Params(C.Structure):
    """Uniforms of the directLighting / pathTracing programs (mirrors orc::Params)."""
    _fields_ = [("sunDirAlt", C.c_float * 4), ("sunEnabled", C.c_int), ("userSphere", C.c_float * 4),
                ("userSphereEm", C.c_float * 3), ("userSphereFlags", C.c_uint32), ("pixelSize", C.c_float),
                ("cameraPos", C.c_float * 3), ("maxSegments", C.c_int), ("minWeight", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("iterations", C.c_uint64), ("nodes", C.c_uint64),
                ("prim_tests", C.c_uint64 * 4), ("segments", C.c_uint64)]

    def algorithmic_bytes(self):
        p = self.prim_tests
        return 48 * self.nodes + 32 * p[0] + 48 * p[1] + 64 * p[2] + 80 * p[3]

    def as_dict(self):
        return dict(rays=self.rays, iterations=self.iterations, nodes=self.nodes,
                    prim_tests=list(self.prim_tests), segments=self.segments,
                    algorithmic_bytes=self.algorithmic_bytes())


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_build_bvh.restype = C.POINTER(C.c_float)
        L.orc_build_bvh.argtypes = [C.POINTER(PrimDesc), C.c_int, C.c_uint, C.c_uint, C.POINTER(C.c_size_t),
                                    C.POINTER(C.c_int)]
        L.orc_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _f4(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _call(name, ins, nout, *extra_before_n):
    ins = [_f4(a) for a in ins]
    n = ins[0].shape[0]
    outs = [np.zeros((n, 4), np.float32) for _ in range(nout)]
    fn = getattr(lib(), name)
    fn(*[_p(a) for a in ins], *extra_before_n, C.c_int(n), *[_p(o) for o in outs])
    return outs


def random(x): return _call("orc_random", [x], 1)[0]
def sincos(x): return _call("orc_sincos", [x], 1)[0]
def pow16(x): return _call("orc_pow16", [x], 1)[0]
def hemisphere(v, ri): return _call("orc_hemisphere", [v, ri], 1)[0]
def inside_cone(v, nrm, ri, half_angle): return _call("orc_inside_cone", [v, nrm, ri], 1, C.c_float(half_angle))[0]
def sphere(rs, rd, sph): return _call("orc_sphere", [rs, rd, sph], 2)
def disc(rs, rd, cr, dn): return _call("orc_disc", [rs, rd, cr, dn], 2)
def triangle(rs, rd, v0, v1, v2): return _call("orc_triangle", [rs, rd, v0, v1, v2], 2)
def cone(rs, rd, q0, q1, q2, q3): return _call("orc_cone", [rs, rd, q0, q1, q2, q3], 2)
def aabb(rs, rd, bmin, bmax): return _call("orc_aabb", [rs, rd, bmin, bmax], 1)[0]


def sky(dirs, sun_dir_alt):
    sda = (C.c_float * 4)(*sun_dir_alt)
    return _call("orc_sky", [dirs], 1, sda)[0]


def traverse(tree, rs, rd, user_sphere=None, want_stats=False):
    tree = np.ascontiguousarray(tree, dtype=np.float32)
    rs, rd = _f4(rs), _f4(rd)
    n = rs.shape[0]
    o0 = np.zeros((n, 4), np.float32)
    o1 = np.zeros((n, 4), np.float32)
    us = (C.c_float * 4)(*user_sphere) if user_sphere is not None else None
    st = Stats()
    lib().orc_traverse(_p(tree), _p(rs), _p(rd), us, C.c_int(n), _p(o0), _p(o1), C.byref(st))
    return (o0, o1, st) if want_stats else (o0, o1)


# ---- host restatement ---------------------------------------------------------------------------
def make_prims(descs):
    """descs: list of (type, [floats])."""
    arr = (PrimDesc * len(descs))()
    for i, (t, f) in enumerate(descs):
        arr[i].type = t
        for k, v in enumerate(f):
            arr[i].f[k] = v
    return arr


def build_bvh(descs, max_levels=1024, min_prims=2):
    """Returns (quads (n,4) float32, max_depth)."""
    arr = descs if isinstance(descs, C.Array) else make_prims(descs)
    nq = C.c_size_t(0)
    md = C.c_int(0)
    p = lib().orc_build_bvh(arr, len(arr), max_levels, min_prims, C.byref(nq), C.byref(md))
    out = np.ctypeslib.as_array(p, shape=(nq.value, 4)).copy()
    lib().orc_free(p)
    return out, md.value


def sort_permutation(keys):
    """std::sort with the reference's centre comparison (src/bvh.cpp:96) on `keys`: original positions in sorted order."""
    k = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(len(k), np.uint32)
    lib().orc_sort_permutation(_p(k), C.c_size_t(len(k)), _p(perm))
    return perm


def camera(pos, dir, up, fov_y, screen_dist, W, H):
    """Returns 13 floats: pos(3) bottomLeft(3) deltaHorz(3) deltaVert(3) pixelSize."""
    out = np.zeros(13, np.float32)
    f3 = lambda v: (C.c_float * 3)(*v)
    lib().orc_camera(f3(pos), f3(dir), f3(up), C.c_float(fov_y), C.c_float(screen_dist), C.c_uint(W), C.c_uint(H), _p(out))
    return out


def sun_direction(az, alt):
    out = (C.c_float * 3)()
    lib().orc_sun_direction(C.c_float(az), C.c_float(alt), out)
    return np.array(list(out), np.float32)


def randseeds(npasses, seed=5489):
    out = np.zeros((npasses, 4), np.float32)
    lib().orc_randseeds(C.c_uint32(seed), C.c_int(npasses), _p(out))
    return out


def cam_rays(cam, W, H):
    rs = np.zeros((H, W, 4), np.float32)
    rd = np.zeros((H, W, 4), np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    lib().orc_cam_rays(_p(cam), W, H, _p(rs), _p(rd))
    return rs, rd


def first_segment_rays(cam, W, H, P, rand_seed, j=0):
    """(rstart, rdir) (H, W, 4) of path j's first segment of every pixel: the jittered camera rays of a path-tracing pass."""
    rs = np.zeros((H, W, 4), np.float32)
    rd = np.zeros((H, W, 4), np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    rsd = (C.c_float * 4)(*[float(x) for x in rand_seed])
    lib().orc_first_segment_rays(_p(cam), W, H, C.byref(P), rsd, C.c_int(j), _p(rs), _p(rd))
    return rs, rd


def pixel_uv(xy, W, H):
    xy = np.ascontiguousarray(xy, np.int32)
    out = np.zeros((xy.shape[0], 2), np.float32)
    lib().orc_pixel_uv(_p(xy), C.c_int(xy.shape[0]), W, H, _p(out))
    return out


def make_params(sun_dir, sun_alt, sun_enabled=True, user_sphere=(0, 0, 0, 0), user_sphere_em=0.0, user_sphere_flags=0,
                pixel_size=0.0, camera_pos=(0, 0, 0), max_segments=5, min_weight=0.01):
    P = Params()
    P.sunDirAlt[:] = [sun_dir[0], sun_dir[1], sun_dir[2], sun_alt]
    P.sunEnabled = 1 if sun_enabled else 0
    P.userSphere[:] = list(user_sphere)
    P.userSphereEm[:] = [user_sphere_em] * 3
    P.userSphereFlags = user_sphere_flags
    P.pixelSize = pixel_size
    P.cameraPos[:] = list(camera_pos)
    P.maxSegments = max_segments
    P.minWeight = min_weight
    return P


def render_direct(tree, cam, W, H, P, tile=None, nthreads=1):
    x0, y0, tw, th = tile or (0, 0, W, H)
    tree = np.ascontiguousarray(tree, np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    out = np.zeros((th, tw, 4), np.float32)
    st = Stats()
    lib().orc_render_direct(_p(tree), _p(cam), W, H, x0, y0, tw, th, C.byref(P), _p(out), nthreads, C.byref(st))
    return out, st


def pt_pass(tree, cam, W, H, P, rand_seed, npaths, accum, tile=None, nthreads=1):
    """In-place: accum (th,tw,4) float32 += pass radiance. Returns Stats."""
    x0, y0, tw, th = tile or (0, 0, W, H)
    tree = np.ascontiguousarray(tree, np.float32)
    cam = np.ascontiguousarray(cam, np.float32)
    assert accum.dtype == np.float32 and accum.shape == (th, tw, 4) and accum.flags.c_contiguous
    rsd = (C.c_float * 4)(*[float(x) for x in rand_seed])
    st = Stats()
    lib().orc_pt_pass(_p(tree), _p(cam), W, H, x0, y0, tw, th, C.byref(P), rsd, npaths, _p(accum), nthreads, C.byref(st))
    return st
