"""ctypes bindings of lib/libgpuart.so (C++ host library, capi.h) and lib/libgpuart_hip.so
(device back end, include/gpuart_hip.h). Plumbing only — no computation happens here.
GPUART_LIBDIR selects another build of the pair: gpuart_amd/lib_test (the product + the test hooks of include/gpuart_hip_test.h:
what tests/conftest.py chooses), or an A/B variant under gpuart_amd/lib_ab/."""
import ctypes as C
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")  # see libgpuart_hip's request_hw_queues(); before any HIP initialisation

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("GPUART_LIBDIR") or os.path.join(HERE, "lib")  # override: A/B runs of differently built libraries
HIP_LIB = os.path.join(LIBDIR, "libgpuart_hip.so")
HOST_LIB = os.path.join(LIBDIR, "libgpuart.so")


class NativeLibraryMissing(RuntimeError):
    pass


class PrimDesc(C.Structure):
    _fields_ = [("type", C.c_int32), ("f", C.c_float * 9)]


class Params(C.Structure):
    _fields_ = [("sunDirAlt", C.c_float * 4), ("sunEnabled", C.c_int32), ("userSphere", C.c_float * 4),
                ("userSphereEm", C.c_float * 3), ("userSphereFlags", C.c_uint32), ("pixelSize", C.c_float),
                ("cameraPos", C.c_float * 3), ("maxSegments", C.c_int32), ("minWeight", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("nodes", C.c_uint64), ("prim_tests", C.c_uint64 * 4), ("segments", C.c_uint64),
                ("box_steps", C.c_uint64), ("box_steps_top", C.c_uint64), ("rewalks", C.c_uint64)]

    def algorithmic_bytes(self):
        """48 B per distinct node + (16 + 16*len) B per tested primitive."""
        p = self.prim_tests
        return 48 * self.nodes + 32 * p[0] + 48 * p[1] + 64 * p[2] + 80 * p[3]

    def as_dict(self):
        return dict(rays=self.rays, nodes=self.nodes, prim_tests=list(self.prim_tests), segments=self.segments,
                    algorithmic_bytes=self.algorithmic_bytes())


_hip = None
_host = None
TEST_LIBDIR = os.path.join(HERE, "lib_test")  # the same sources + the hooks of include/gpuart_hip_test.h (csrc/Makefile)


class _HipLibrary(C.CDLL):
    """libgpuart_hip.so; says where the test hooks live when the product library is asked for one."""

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith("gpuart_hip_test_"):
                raise AttributeError("%s is a test hook (include/gpuart_hip_test.h): the product library %s has none — load the test build with "
                                     "GPUART_LIBDIR=%s" % (name, self._name, TEST_LIBDIR)) from None
            raise


def hip_lib():
    """libgpuart_hip.so; raises NativeLibraryMissing if it has not been built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise NativeLibraryMissing("%s not found — run `python -c 'import __graft_entry__ as g; g.build()'` "
                                       "(make -C gpuart_amd/csrc); there is no CPU fallback" % HIP_LIB)
        L = _HipLibrary(HIP_LIB, mode=C.RTLD_GLOBAL)
        L.gpuart_hip_last_error.restype = C.c_char_p
        L.gpuart_hip_frame_row.restype = C.c_uint32
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB):
            raise NativeLibraryMissing("%s not found — run make -C gpuart_amd/csrc" % HOST_LIB)
        L = C.CDLL(HOST_LIB)
        L.gpuart_renderer_create.restype = C.c_void_p
        L.gpuart_renderer_backend.restype = C.c_void_p
        L.gpuart_renderer_path_tracing_pass.restype = C.c_uint
        for name in ["destroy", "is_ok", "set_primitives", "init_box", "init_dragon", "init_cluster", "init_tree", "set_camera", "update_viewport",
                     "set_tile", "set_interleaved_tile", "set_sun", "set_user_sphere", "set_max_path_segments", "set_seed", "render_direct",
                     "restart_path_tracing", "path_tracing_pass", "read_direct", "read_radiance", "finish", "backend",
                     "save_checkpoint", "load_checkpoint",
                     "params", "scene_info"]:
            getattr(L, "gpuart_renderer_" + name).argtypes = None
        _host = L
    return _host


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def make_prims(descs):
    arr = (PrimDesc * len(descs))()
    for i, (t, f) in enumerate(descs):
        arr[i].type = t
        for k, v in enumerate(f):
            arr[i].f[k] = v
    return arr


# ---- pure host functions ------------------------------------------------------------------------
def compile_bvh(descs, max_levels=1024, min_prims=2):
    """BoundingVolumesHierarchy(prims, max_levels, min_prims).Compile() -> (quads (n,4) float32, depth)."""
    L = host_lib()
    arr = descs if isinstance(descs, C.Array) else make_prims(descs)
    q = C.POINTER(C.c_float)()
    nq = C.c_size_t(0)
    depth = C.c_uint(0)
    rc = L.gpuart_compile_bvh(arr, C.c_int(len(arr)), C.c_uint(max_levels), C.c_uint(min_prims), C.byref(q), C.byref(nq),
                              C.byref(depth))
    if rc != 0:
        raise RuntimeError("gpuart_compile_bvh failed")
    out = np.ctypeslib.as_array(q, shape=(nq.value, 4)).copy()
    L.gpuart_free(q)
    return out, depth.value


def last_build_ms():
    """(build ms, compile ms) the host library itself spent in the last compile_bvh / compile_bvh_from_file of this process."""
    out = (C.c_double * 2)()
    host_lib().gpuart_last_build_ms(out)
    return float(out[0]), float(out[1])


def compile_bvh_from_file(kind, path, magnification=1.0, translation=(0, 0, 0), extra=()):
    """kind: 'ply' or 'dat'. Returns (quads, depth, nloaded)."""
    L = host_lib()
    ex = make_prims(list(extra))
    q = C.POINTER(C.c_float)()
    nq = C.c_size_t(0)
    depth = C.c_uint(0)
    nl = C.c_size_t(0)
    rc = L.gpuart_compile_bvh_from_file(C.c_int(0 if kind == "ply" else 1), path.encode(), C.c_float(magnification),
                                        _f3(translation), ex, C.c_int(len(ex)), C.byref(q), C.byref(nq), C.byref(depth),
                                        C.byref(nl))
    if rc != 0:
        raise RuntimeError("loading %s failed" % path)
    out = np.ctypeslib.as_array(q, shape=(nq.value, 4)).copy()
    L.gpuart_free(q)
    return out, depth.value, nl.value


def sort_permutation(keys, threads=8):
    """gpuart_sort_permutation: the BVH build's parallel exact sort (csrc/host/exact_sort.h) on `keys`."""
    k = np.ascontiguousarray(keys, np.float32)
    perm = np.zeros(len(k), np.uint32)
    host_lib().gpuart_sort_permutation(_p(k), C.c_size_t(len(k)), C.c_uint(threads), _p(perm))
    return perm


def camera_basis(pos, dir, up, fov_y, screen_dist, W, H):
    out = np.zeros(13, np.float32)
    host_lib().gpuart_camera_basis(_f3(pos), _f3(dir), _f3(up), C.c_float(fov_y), C.c_float(screen_dist), C.c_uint(W),
                                   C.c_uint(H), _p(out))
    return out


def sun_direction(az, alt):
    out = (C.c_float * 3)()
    host_lib().gpuart_sun_direction(C.c_float(az), C.c_float(alt), out)
    return np.array(list(out), np.float32)


def vec3_ops(a, b, s, dtype=np.float32):
    """Every Vec3 operation of csrc/host/math_types.h once (36 values; gpuart_vec3f_ops / gpuart_vec3d_ops)."""
    a = np.ascontiguousarray(a, dtype); b = np.ascontiguousarray(b, dtype)
    out = np.zeros(36, dtype)
    if dtype == np.float32:
        host_lib().gpuart_vec3f_ops(_p(a), _p(b), C.c_float(float(s)), _p(out))
    else:
        host_lib().gpuart_vec3d_ops(_p(a), _p(b), C.c_double(float(s)), _p(out))
    return out


# ---- device back end (include/gpuart_hip.h) -----------------------------------------------------------
class TileGeom(C.Structure):
    """gpuart_tile_geom (include/gpuart_hip.h): one rank's share of a frame sharded by rows."""
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("x0", C.c_uint32), ("y0", C.c_uint32), ("tw", C.c_uint32),
                ("th", C.c_uint32), ("band_rows", C.c_uint32), ("band_stride", C.c_uint32)]

    def rows(self):
        """Frame rows of the share, in local row order."""
        L = hip_lib()
        return np.array([L.gpuart_hip_frame_row(C.byref(self), C.c_uint32(k)) for k in range(self.th)], np.int64)


def share_of_rank(W, H, rank, nranks, band_rows=8):
    """gpuart_hip_share_of_rank: bands of band_rows rows dealt round-robin to the ranks (pure host code)."""
    g = TileGeom()
    rc = hip_lib().gpuart_hip_share_of_rank(C.c_uint32(W), C.c_uint32(H), rank, nranks, C.c_uint32(band_rows), C.byref(g))
    if rc:
        raise ValueError("gpuart_hip_share_of_rank(%d, %d, %d, %d, %d) -> %d" % (W, H, rank, nranks, band_rows, rc))
    return g


def scatter_rows_host(g, tile_rgba, full_rgba):
    """gpuart_hip_scatter_rows_host: rows of a share (th x tw x 4) into the full frame (H x W x 4), on the host."""
    tile = np.ascontiguousarray(tile_rgba, np.float32)
    assert tile.shape == (g.th, g.tw, 4) and full_rgba.shape == (g.H, g.W, 4) and full_rgba.flags["C_CONTIGUOUS"]
    rc = hip_lib().gpuart_hip_scatter_rows_host(C.byref(g), _p(tile), _p(full_rgba))
    if rc:
        raise ValueError("gpuart_hip_scatter_rows_host -> %d" % rc)
    return full_rgba


def tree_class(quads):
    """What the uploader decides about a compiled tree, without a device: dict(irregular, disorderly, type_mask); the fast kernels
    walk nearest-child-first iff neither flag is set."""
    L = hip_lib()
    q = np.ascontiguousarray(quads, np.float32)
    flags = C.c_uint32(0)
    rc = L.gpuart_hip_test_tree_class(_p(q), C.c_size_t(q.size // 4), C.byref(flags))
    if rc != 0:
        raise HipError("gpuart_hip error %d: %s" % (rc, L.gpuart_hip_last_error().decode()))
    return dict(irregular=bool(flags.value & 1), disorderly=bool(flags.value & 2), type_mask=(flags.value >> 8) & 15)


def tree_slack(quads):
    """(slack constant the upload would give the tree for its quick box answers — inf: none —, a box plane is subnormal), without a device."""
    L = hip_lib()
    q = np.ascontiguousarray(quads, np.float32)
    slack, flags = C.c_float(0), C.c_uint32(0)
    for rc in (L.gpuart_hip_test_tree_slack(_p(q), C.c_size_t(q.size // 4), C.byref(slack)),
               L.gpuart_hip_test_tree_class(_p(q), C.c_size_t(q.size // 4), C.byref(flags))):
        if rc != 0:
            raise HipError("gpuart_hip error %d: %s" % (rc, L.gpuart_hip_last_error().decode()))
    return float(slack.value), bool(flags.value & 4)


def comm_library():
    """Path of the RCCL library libgpuart_hip resolved its entry points from (dladdr of ncclCommInitRank)."""
    L = hip_lib()
    buf = C.create_string_buffer(4096)
    rc = L.gpuart_hip_comm_library(buf, C.c_size_t(len(buf)))
    if rc != 0:
        raise HipError("gpuart_hip error %d: %s" % (rc, L.gpuart_hip_last_error().decode()))
    return buf.value.decode()


def comm_unique_id():
    """gpuart_hip_comm_unique_id (ncclGetUniqueId): 128 bytes for rank 0 to hand to the other ranks."""
    buf = (C.c_ubyte * 128)()
    L = hip_lib()
    if L.gpuart_hip_comm_unique_id(buf):
        raise HipError(L.gpuart_hip_last_error().decode())
    return bytes(buf)


def phase_begin(name, timeout_ms):
    """gpuart_hip_phase_begin: names the phase the process enters and arms the library's watchdog (a native thread: it fires
    whatever the interpreter is doing) — a phase that outlives timeout_ms ends the process with exit code WATCHDOG_EXIT after
    printing the phase and the library's recent errors."""
    hip_lib().gpuart_hip_phase_begin(str(name).encode(), C.c_uint32(int(timeout_ms)))


def phase_end():
    hip_lib().gpuart_hip_phase_end()


def phase_log(on):
    """gpuart_hip_phase_log: phase lines on / off from now on (the watchdog stays armed either way); returns the previous setting."""
    return bool(hip_lib().gpuart_hip_phase_log(C.c_int(1 if on else 0)))


class phase:
    """with phase("communicator init", 120000): ...   (the phase ends when the block does, exception or not)"""

    def __init__(self, name, timeout_ms):
        self.name, self.timeout_ms = name, timeout_ms

    def __enter__(self):
        phase_begin(self.name, self.timeout_ms)

    def __exit__(self, *exc):
        phase_end()
        return False


WATCHDOG_EXIT = 86
ERR_TIMEOUT = -4


def comm_stuck():
    """True once an RCCL call of this process has outlived GPUART_HIP_COMM_TIMEOUT_MS (its thread is parked in it)."""
    return bool(hip_lib().gpuart_hip_comm_stuck())


def comm_init_all(backends):
    """gpuart_hip_comm_init_all: the contexts become the ranks 0..n-1 of one communicator (ncclCommInitAll)."""
    L = hip_lib()
    arr = (C.c_void_p * len(backends))(*[b.ctx for b in backends])
    backends[0]._chk(L.gpuart_hip_comm_init_all(arr, len(backends)))


def gather_all_read(backends, which, divide_by, root, W, H):
    """gpuart_hip_gather_all_read: the frame assembled on the root's device from every context's share, read back (H, W, 4)."""
    L = hip_lib()
    arr = (C.c_void_p * len(backends))(*[b.ctx for b in backends])
    out = np.empty((H, W, 4), np.float32)
    backends[0]._chk(L.gpuart_hip_gather_all_read(arr, len(backends), which, C.c_float(divide_by), root, _p(out)))
    return out


class HipError(RuntimeError):
    """A gpuart_hip_* call returned an error; `code` is the library's (GPUART_HIP_ERR_*: -4 a bounded wait ran out)."""
    code = None


class Backend:
    """A gpuart_hip_ctx. Owns the context unless constructed from a borrowed pointer."""

    def __init__(self, device=0, borrowed=None):
        self.L = hip_lib()
        self.owned = borrowed is None
        if borrowed is None:
            ctx = C.c_void_p()
            self._chk(self.L.gpuart_hip_create(C.c_int(device), C.byref(ctx)))
            self.ctx = ctx
        else:
            self.ctx = C.c_void_p(borrowed)
        self.tile = None

    def _chk(self, rc):
        if rc != 0:
            e = HipError("gpuart_hip error %d: %s" % (rc, self.L.gpuart_hip_last_error().decode()))
            e.code = rc
            raise e

    def close(self):
        if self.owned and self.ctx:
            self.L.gpuart_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # frame / scene / camera
    def resize(self, W, H):
        self._chk(self.L.gpuart_hip_resize(self.ctx, C.c_uint32(W), C.c_uint32(H)))
        self.tile = (0, 0, W, H)

    def set_tile(self, x0, y0, tw, th):
        self._chk(self.L.gpuart_hip_set_tile(self.ctx, C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(tw), C.c_uint32(th)))
        self.tile = (x0, y0, tw, th)

    def set_tile_interleaved(self, x0, y0, tw, th_local, band_rows, band_stride):
        self._chk(self.L.gpuart_hip_set_tile_interleaved(self.ctx, C.c_uint32(x0), C.c_uint32(y0), C.c_uint32(tw), C.c_uint32(th_local),
                                                         C.c_uint32(band_rows), C.c_uint32(band_stride)))
        self.tile = (x0, y0, tw, th_local)

    def upload_bvh(self, quads):
        quads = np.ascontiguousarray(quads, np.float32)
        self._chk(self.L.gpuart_hip_upload_bvh(self.ctx, _p(quads), C.c_size_t(quads.size // 4)))

    def set_camera(self, cam):
        cam = np.ascontiguousarray(cam, np.float32)
        self._chk(self.L.gpuart_hip_set_camera(self.ctx, _f3(cam[0:3]), _f3(cam[3:6]), _f3(cam[6:9]), _f3(cam[9:12])))

    # rendering
    def render_direct(self, params):
        self._chk(self.L.gpuart_hip_render_direct(self.ctx, C.byref(params)))

    def pt_reset(self):
        self._chk(self.L.gpuart_hip_pt_reset(self.ctx))

    def pt_plan(self, passes):
        self._chk(self.L.gpuart_hip_pt_plan(self.ctx, C.c_uint32(passes)))

    def flush(self):
        self._chk(self.L.gpuart_hip_flush(self.ctx))

    def pt_pass(self, params, rand_seed, npaths):
        rs = (C.c_float * 4)(*[float(x) for x in rand_seed])
        self._chk(self.L.gpuart_hip_pt_pass(self.ctx, C.byref(params), rs, C.c_int(npaths)))

    def read(self, which, divide_by=1.0):
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.float32)
        self._chk(self.L.gpuart_hip_read(self.ctx, C.c_int(which), _p(out), C.c_float(divide_by)))
        return out

    def write(self, which, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        self._chk(self.L.gpuart_hip_write(self.ctx, C.c_int(which), _p(rgba)))

    def export(self, which, device_ptr, divide_by=1.0):
        self._chk(self.L.gpuart_hip_export(self.ctx, C.c_int(which), C.c_void_p(device_ptr), C.c_float(divide_by)))

    def set_share(self, g):
        self._chk(self.L.gpuart_hip_set_share(self.ctx, C.byref(g)))
        self.tile = (g.x0, g.y0, g.tw, g.th)

    def get_share(self):
        g = TileGeom()
        self._chk(self.L.gpuart_hip_get_share(self.ctx, C.byref(g)))
        return g

    def comm_init(self, nranks, rank, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        self._chk(self.L.gpuart_hip_comm_init(self.ctx, nranks, rank, buf))

    def comm_destroy(self):
        self._chk(self.L.gpuart_hip_comm_destroy(self.ctx))

    def gather(self, which, divide_by, root, full_frame_device_ptr):
        """Collective: assembles every rank's share on `root` (full_frame_device_ptr: W*H*4 floats on the root, else 0)."""
        self._chk(self.L.gpuart_hip_gather(self.ctx, which, C.c_float(divide_by), root, C.c_void_p(full_frame_device_ptr or None)))

    def comm_info(self):
        """(ranks, own rank) as the communicator itself reports them (ncclCommCount, ncclCommUserRank)."""
        n, me = C.c_int(0), C.c_int(0)
        self._chk(self.L.gpuart_hip_comm_info(self.ctx, C.byref(n), C.byref(me)))
        return n.value, me.value

    def finish(self):
        self._chk(self.L.gpuart_hip_finish(self.ctx))

    def wait(self, timeout_ms):
        """finish() with a bound: HipError with code ERR_TIMEOUT if the context's work is not complete after timeout_ms."""
        self._chk(self.L.gpuart_hip_wait(self.ctx, C.c_uint32(int(timeout_ms))))

    def test_tile_order(self, order):
        """Birth order of the paths of a pass: a permutation of the tile's 8x8 pixel blocks (None: row-major)."""
        if order is None:
            self._chk(self.L.gpuart_hip_test_tile_order(self.ctx, None, C.c_size_t(0)))
            return
        o = np.ascontiguousarray(order, np.uint32)
        self._chk(self.L.gpuart_hip_test_tile_order(self.ctx, o.ctypes.data_as(C.c_void_p), C.c_size_t(o.size)))

    def test_current_tile_order(self):
        """The birth order in use (None: row-major)."""
        g = self.get_share()
        out = np.empty(((g.tw + 7) // 8) * ((g.th + 7) // 8), np.uint32)
        rc = self.L.gpuart_hip_test_current_tile_order(self.ctx, out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size))
        if rc < 0:
            self._chk(rc)
        return out if rc == 1 else None

    def test_sort_tiles(self, cost):
        """k_tile_order alone: per-block cost -> birth order (most expensive class first, row-major within a class)."""
        cst = np.ascontiguousarray(cost, np.uint32)
        out = np.empty(cst.size, np.uint32)
        self._chk(self.L.gpuart_hip_test_sort_tiles(self.ctx, cst.ctypes.data_as(C.c_void_p), C.c_size_t(cst.size), out.ctypes.data_as(C.c_void_p)))
        return out

    def test_stall(self, ms):
        """Keeps the context's primary stream busy for `ms` milliseconds (what a missing peer looks like to the bounded waits)."""
        self._chk(self.L.gpuart_hip_test_stall(self.ctx, C.c_uint32(int(ms))))

    MODE_WAVEFRONT, MODE_REFERENCE_WORK, MODE_MEGAKERNEL = 0, 1, 2

    def set_mode(self, mode):
        """0 wavefront (fast, default), 1 reference-work (exact counters), 2 megakernel."""
        self._chk(self.L.gpuart_hip_set_mode(self.ctx, C.c_int(int(mode))))

    def set_timing(self, level):
        self._chk(self.L.gpuart_hip_set_timing(self.ctx, C.c_int(int(level))))

    def counters(self, reset=False):
        c = Counters()
        self._chk(self.L.gpuart_hip_counters(self.ctx, C.byref(c), C.c_int(1 if reset else 0)))
        return c

    def kernel_time(self, cls=0, reset=False):
        """(total ms, count) of HIP-event-timed intervals: cls 0 = render calls, 1 = BVH-query kernels."""
        ms = C.c_double(0)
        n = C.c_uint64(0)
        self._chk(self.L.gpuart_hip_kernel_time(self.ctx, C.c_int(cls), C.byref(ms), C.byref(n), C.c_int(1 if reset else 0)))
        return ms.value, n.value

    def scene_info(self):
        nodes, prims, bytes_ = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        depth = C.c_uint32(0)
        self._chk(self.L.gpuart_hip_scene_info(self.ctx, C.byref(nodes), C.byref(prims), C.byref(depth), C.byref(bytes_)))
        return dict(nodes=nodes.value, prims=prims.value, max_depth=depth.value, device_bytes=bytes_.value)

    def set_nearest_first(self, min_prims):
        """gpuart_hip_set_nearest_first: trees of at least min_prims primitives are walked nearer child first (opt-in; 0xffffffff: never)."""
        self._chk(self.L.gpuart_hip_set_nearest_first(self.ctx, C.c_uint32(int(min_prims))))

    def scene_order(self):
        """0: nearer child first (certified); 1: the reference's order (small tree); 2: the reference's order, exact box tests."""
        o = C.c_int(-1)
        self._chk(self.L.gpuart_hip_scene_order(self.ctx, C.byref(o)))
        return o.value

    # test hooks
    def _hook(self, name, ins, nout, *extra):
        ins = [np.ascontiguousarray(a, np.float32) for a in ins]
        n = ins[0].shape[0]
        outs = [np.zeros((n, 4), np.float32) for _ in range(nout)]
        self._chk(getattr(self.L, name)(self.ctx, *[_p(a) for a in ins], *extra, C.c_int(n), *[_p(o) for o in outs]))
        return outs

    def test_random(self, x): return self._hook("gpuart_hip_test_random", [x], 1)[0]
    def test_math(self, x): return self._hook("gpuart_hip_test_math", [x], 1)[0]
    def test_hemisphere(self, v, ri): return self._hook("gpuart_hip_test_hemisphere", [v, ri], 1)[0]

    def test_inside_cone(self, v, nrm, ri, half_angle):
        return self._hook("gpuart_hip_test_inside_cone", [v, nrm, ri], 1, C.c_float(half_angle))[0]

    def test_sky(self, dirs, sun_dir_alt):
        return self._hook("gpuart_hip_test_sky", [dirs], 1, (C.c_float * 4)(*[float(x) for x in sun_dir_alt]))[0]

    def test_intersect(self, ptype, rs, rd, quads16):
        rs = np.ascontiguousarray(rs, np.float32)
        rd = np.ascontiguousarray(rd, np.float32)
        q = np.ascontiguousarray(quads16, np.float32).reshape(rs.shape[0], 16)
        n = rs.shape[0]
        o0, o1 = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
        self._chk(self.L.gpuart_hip_test_intersect(self.ctx, C.c_int(ptype), _p(rs), _p(rd), _p(q), C.c_int(n), _p(o0), _p(o1)))
        return o0, o1

    def test_aabb(self, rs, rd, bmin, bmax): return self._hook("gpuart_hip_test_aabb", [rs, rd, bmin, bmax], 1)[0]

    def test_traverse(self, rs, rd, user_sphere, any_hit=False, nearest_first=False):
        """any_hit: only "anything hit?"; nearest_first: the closest-hit query in the order of the fast kernels (nearer child first)."""
        rs = np.ascontiguousarray(rs, np.float32)
        rd = np.ascontiguousarray(rd, np.float32)
        n = rs.shape[0]
        o0, o1 = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
        us = (C.c_float * 4)(*[float(x) for x in user_sphere])
        self._chk(self.L.gpuart_hip_test_traverse(self.ctx, _p(rs), _p(rd), us, C.c_int(n), C.c_int(2 if nearest_first else 1 if any_hit else 0),
                                                  _p(o0), _p(o1)))
        return o0, o1

    def test_cam_rays(self):
        _, _, tw, th = self.tile
        rs, rd = np.zeros((th, tw, 4), np.float32), np.zeros((th, tw, 4), np.float32)
        self._chk(self.L.gpuart_hip_test_cam_rays(self.ctx, _p(rs), _p(rd)))
        return rs, rd


# ---- gpuart::Renderer ------------------------------------------------------------------------------
class Renderer:
    """Python handle of the C++ gpuart::Renderer (reference API, src/renderer.h:183-296)."""

    def __init__(self, W, H, cam, device=0):
        self.L = host_lib()
        self.W, self.H = W, H
        self.tile = (0, 0, W, H)
        self.h = C.c_void_p(self.L.gpuart_renderer_create(C.c_uint(W), C.c_uint(H), _f3(cam["pos"]), _f3(cam["dir"]),
                                                          _f3(cam["up"]), C.c_float(cam["fov_y"]),
                                                          C.c_float(cam["screen_dist"]), C.c_int(device)))
        if not self.L.gpuart_renderer_is_ok(self.h):
            msg = hip_lib().gpuart_hip_last_error().decode()
            self.L.gpuart_renderer_destroy(self.h)
            self.h = None
            raise HipError("Renderer initialisation failed: " + msg)
        self.backend = Backend(borrowed=self.L.gpuart_renderer_backend(self.h))
        self.backend.tile = self.tile

    def close(self):
        if self.h:
            self.L.gpuart_renderer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def is_ok(self): return bool(self.L.gpuart_renderer_is_ok(self.h))

    def set_primitives(self, descs, print_info=False):
        arr = descs if isinstance(descs, C.Array) else make_prims(descs)
        self.L.gpuart_renderer_set_primitives(self.h, arr, C.c_int(len(arr)), C.c_int(1 if print_info else 0))

    def init_box(self): self.L.gpuart_renderer_init_box(self.h)
    def init_dragon(self, path): return bool(self.L.gpuart_renderer_init_dragon(self.h, path.encode()))
    def init_cluster(self, path): return bool(self.L.gpuart_renderer_init_cluster(self.h, path.encode()))
    def init_tree(self, path): return bool(self.L.gpuart_renderer_init_tree(self.h, path.encode()))

    def set_camera(self, cam):
        return bool(self.L.gpuart_renderer_set_camera(self.h, _f3(cam["pos"]), _f3(cam["dir"]), _f3(cam["up"]),
                                                      C.c_float(cam["fov_y"]), C.c_float(cam["screen_dist"])))

    def update_viewport(self, W, H):
        ok = bool(self.L.gpuart_renderer_update_viewport(self.h, C.c_uint(W), C.c_uint(H)))
        self.W, self.H = W, H
        self.tile = self.backend.tile = (0, 0, W, H)
        return ok

    def set_tile(self, x0, y0, w, h):
        ok = bool(self.L.gpuart_renderer_set_tile(self.h, C.c_uint(x0), C.c_uint(y0), C.c_uint(w), C.c_uint(h)))
        if ok:
            self.tile = self.backend.tile = (x0, y0, w, h)
        return ok

    def set_interleaved_tile(self, x0, y0, w, local_rows, band_rows, band_stride):
        ok = bool(self.L.gpuart_renderer_set_interleaved_tile(self.h, C.c_uint(x0), C.c_uint(y0), C.c_uint(w), C.c_uint(local_rows),
                                                              C.c_uint(band_rows), C.c_uint(band_stride)))
        if ok:
            self.tile = self.backend.tile = (x0, y0, w, local_rows)
        return ok

    def set_sun(self, az, alt, direct=True):
        self.L.gpuart_renderer_set_sun(self.h, C.c_float(az), C.c_float(alt), C.c_int(1 if direct else 0))

    def set_user_sphere(self, pos, radius, emittance=0.0, specular=False, fuzzy=False):
        self.L.gpuart_renderer_set_user_sphere(self.h, _f3(pos), C.c_float(radius), C.c_float(emittance),
                                               C.c_int(int(specular)), C.c_int(int(fuzzy)))

    def set_max_path_segments(self, n): self.L.gpuart_renderer_set_max_path_segments(self.h, C.c_uint(n))
    def set_seed(self, seed): self.L.gpuart_renderer_set_seed(self.h, C.c_uint32(seed))
    def render_direct(self): self.L.gpuart_renderer_render_direct(self.h)

    def restart_path_tracing(self, per_pass, per_pixel):
        self.L.gpuart_renderer_restart_path_tracing(self.h, C.c_uint(per_pass), C.c_uint(per_pixel))

    def path_tracing_pass(self): return int(self.L.gpuart_renderer_path_tracing_pass(self.h))

    def read_direct(self):
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.float32)
        if not self.L.gpuart_renderer_read_direct(self.h, _p(out)):
            raise HipError("read_direct failed")
        return out

    def read_radiance(self, normalized=False):
        _, _, tw, th = self.tile
        out = np.empty((th, tw, 4), np.float32)
        if not self.L.gpuart_renderer_read_radiance(self.h, _p(out), C.c_int(1 if normalized else 0)):
            raise HipError("read_radiance failed")
        return out

    def finish(self): return bool(self.L.gpuart_renderer_finish(self.h))

    def last_setprims_ms(self):
        """(whole call, BVH build, compilation, re-layout + upload) of the last set_primitives, ms, as the library timed them."""
        out = (C.c_double * 4)()
        self.L.gpuart_renderer_last_setprims_ms(self.h, out)
        return tuple(float(x) for x in out)

    def save_checkpoint(self, path): return bool(self.L.gpuart_renderer_save_checkpoint(self.h, path.encode()))
    def load_checkpoint(self, path): return bool(self.L.gpuart_renderer_load_checkpoint(self.h, path.encode()))

    def params(self):
        p = Params()
        self.L.gpuart_renderer_params(self.h, C.byref(p))
        return p
