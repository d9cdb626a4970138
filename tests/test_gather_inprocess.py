"""The N > 1 branches of the multi-GPU read-out, executed on ONE GPU.

The step: the reference normalises the accumulated radiance into the default framebuffer (src/renderer.cpp:601-616,
shaders/pt_normalize.glsl:44-47); for a frame whose rows live on several GPUs the library gathers them on a root
(gpuart_hip.hip: gather_post — the root's receive offsets, the peers' sends —, gather_place — the d_stage + off sources —,
gpuart_hip_gather_all over several contexts, gpuart_hip_gather's share exchange). Until round 6 all of that had run with one rank only.

Here N = 2, 3, 8 contexts live on device 0 and talk through tests/stubs/rccl_stub.cpp, an in-process stand-in for librccl.so that serves
ncclSend / ncclRecv / ncclAllGather as stream-ordered device-to-device copies (GPUART_HIP_RCCL_LIBRARY; GPUART_HIP_TEST_SHARED_DEVICE=1
lets gpuart_hip_comm_init_all put two ranks on one device — a real RCCL refuses that by itself). Every gathered frame must equal the frame
ONE context renders, bit for bit: roots 0 and N-1, a ragged last band, an empty share (more ranks than bands), the one-thread form
(_comm_init_all + _gather_all_read, what gpuart_cli --gpus N calls) and the one-thread-per-rank form (_comm_init + _gather, what bench.py's
ranks call), and gpuart_cli --gpus N itself.

Each case runs in a child process: the library resolves its RCCL once per process, from the environment."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CLI = os.path.join(ROOT, "gpuart_amd", "bin", "gpuart_cli")

pytestmark = [pytest.mark.gpu, pytest.mark.rccl]


def _child(code, stub, timeout=300, **more):
    env = dict(os.environ, GPUART_HIP_RCCL_LIBRARY=stub, GPUART_HIP_TEST_SHARED_DEVICE="1", GPUART_HIP_COMM_TIMEOUT_MS="60000",
               GPUART_HIP_GATHER_TIMEOUT_MS="60000", GPUART_HIP_PHASE_LOG="0", **more)
    return subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], capture_output=True, text=True, env=env,
                          timeout=timeout)


_SETUP = """
import ctypes as C, threading, numpy as np
from gpuart_amd import binding as B
from gpuart_amd import synth_scenes as S
stub = C.CDLL(B.comm_library())
assert B.comm_library().endswith("librccl_stub.so"), B.comm_library()
def served():
    out = (C.c_uint64 * 3)()
    stub.rccl_stub_served(out)
    return tuple(int(x) for x in out)
cam = dict(S.DEFAULT_CAMERA); cam["dir"] = S.camera_dir(cam)

def renderer(W, H, share=None):
    r = B.Renderer(W, H, cam, device=0)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.25, 2.0)
    r.init_box()
    assert r.is_ok()
    if share is not None:
        g = B.share_of_rank(W, H, share[0], share[1])
        if g.th:
            assert r.set_interleaved_tile(g.x0, g.y0, g.tw, g.th, g.band_rows, g.band_stride)
        else:
            r.backend.set_share(g)          # an empty share (more ranks than bands): through the C ABI, the Renderer refuses to be one
    return r

def render(r, K):
    r.restart_path_tracing(1, K)
    for _ in range(K):
        r.path_tracing_pass()

def same(a, b):
    return bool((a.view(np.uint32) == b.view(np.uint32)).all())
"""


@pytest.mark.parametrize("N", [2, 3, 8])
def test_frames_gathered_from_n_contexts_equal_the_frame_of_one(rccl_stub, N):
    """_comm_init_all + _gather_all_read, one thread driving every rank (gpuart_cli --gpus N). 100 x 52: seven bands, the last one ragged
    (4 rows); with N = 8 the eighth rank's share is empty. Roots 0 and N-1 (the root's own share first / last in the staging area), two
    gathers per communicator (staging and send buffers reused), the raw accumulator and the normalised one."""
    p = _child(_SETUP + """
N, W, H, K = %d, 100, 52, 3
one = renderer(W, H); render(one, K)
want_sum, want = one.read_radiance(False), one.read_radiance(True)
ranks = [renderer(W, H, (k, N)) for k in range(N)]
shares = [B.share_of_rank(W, H, k, N) for k in range(N)]
assert sum(g.th for g in shares) == H and (N < 8 or shares[7].th == 0) and min(g.th for g in shares[:7]) >= 4
for r in ranks: render(r, K)
bes = [r.backend for r in ranks]
B.comm_init_all(bes)
for k, b in enumerate(bes): assert b.comm_info() == (N, k)
senders = lambda root: sum(1 for k, g in enumerate(shares) if k != root and g.th)
before = served()
for root in (0, N - 1):
    full = B.gather_all_read(bes, 1, float(K), root, W, H)
    assert same(full, want), "root %%d: %%d pixels differ" %% (root, int((full.view(np.uint32) != want.view(np.uint32)).any(-1).sum()))
    full = B.gather_all_read(bes, 1, 1.0, root, W, H)
    assert same(full, want_sum), "root %%d, raw accumulator" %% root
after = served()
assert after[0] - before[0] == 2 * (senders(0) + senders(N - 1)), (before, after)      # the transfers really happened, one per non-empty peer
assert after[1] - before[1] == 2 * 16 * W * sum(g.th for k, g in enumerate(shares) if k != 0) + 2 * 16 * W * sum(g.th for k, g in enumerate(shares) if k != N - 1)
# more passes on top, gathered again: the communicator and the buffers are still good
for r in ranks + [one]:
    r.restart_path_tracing(1, K + 2)
    for _ in range(K + 2): r.path_tracing_pass()
full = B.gather_all_read(bes, 1, float(K + 2), 1 %% N, W, H)
assert same(full, one.read_radiance(True))
for b in bes: b.comm_destroy()
assert not B.comm_stuck()
print("gathered", after[0] - before[0])
""" % N, rccl_stub)
    assert p.returncode == 0 and "gathered" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


def test_the_bench_shape_1080p_over_eight_ranks(rccl_stub):
    """bench.py --gpus 8 as far as one GPU can run it: the 1920x1080 frame of Scene D's camera on the box scene, 8 shares of 17 / 16 bands (135
    bands of 8 rows), two passes, gathered on rank 0 through the library's transfers — equal to the single-context frame bit for bit, 7
    transfers of the peers' 135 or 128 rows each. (Two pass lanes and runs of at most 4 passes per context: nine contexts share one device.)"""
    p = _child(_SETUP + """
N, W, H, K = 8, 1920, 1080, 2
one = renderer(W, H); render(one, K)
want = one.read_radiance(True)
ranks = [renderer(W, H, (k, N)) for k in range(N)]
shares = [B.share_of_rank(W, H, k, N) for k in range(N)]
assert [g.th for g in shares] == [136] * 7 + [128] and sum(g.th for g in shares) == H
for r in ranks: render(r, K)
bes = [r.backend for r in ranks]
B.comm_init_all(bes)
before = served()
full = B.gather_all_read(bes, 1, float(K), 0, W, H)
after = served()
assert same(full, want), "%d pixels differ" % int((full.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
assert after[0] - before[0] == 7 and after[1] - before[1] == 16 * W * (H - 136), (before, after)
for b in bes: b.comm_destroy()
print("gathered 1080p")
""", rccl_stub, GPUART_HIP_PASSES_IN_FLIGHT="2", GPUART_HIP_MAX_BATCH="4")
    assert p.returncode == 0 and "gathered 1080p" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


def test_cfg5_decomposition_7680x4320_over_eight_ranks(rccl_stub):
    """BASELINE.json's cfg5 as a decomposition: the 7680x4320 frame as eight interleaved shares of 68 / 67 bands (540 bands of 8 rows), gathered on
    rank 0 through the library's transfers (seven of 66 MB each, 531 MB assembled) — equal to the frame one context renders, bit for bit. One
    pass of the box scene: the 256 passes per pixel of the configuration are covered on a share by test_cfg4_and_cfg5_at_their_stated_depth,
    the scaling by profiles/r06/tile_scaling_one_gpu_cfg5.txt; what remains for an 8-GPU node is RCCL itself."""
    p = _child(_SETUP + """
N, W, H, K = 8, 7680, 4320, 1
one = renderer(W, H); render(one, K)
want = one.read_radiance(True)
one.close()
ranks = [renderer(W, H, (k, N)) for k in range(N)]
shares = [B.share_of_rank(W, H, k, N) for k in range(N)]
assert [g.th for g in shares] == [544] * 4 + [536] * 4 and sum(g.th for g in shares) == H
for r in ranks: render(r, K)
bes = [r.backend for r in ranks]
B.comm_init_all(bes)
before = served()
full = B.gather_all_read(bes, 1, float(K), 0, W, H)
after = served()
assert same(full, want), "%d pixels differ" % int((full.view(np.uint32) != want.view(np.uint32)).any(-1).sum())
assert after[0] - before[0] == 7 and after[1] - before[1] == 16 * W * (H - 544), (before, after)
for b in bes: b.comm_destroy()
print("gathered 8K")
""", rccl_stub, timeout=600, GPUART_HIP_PASSES_IN_FLIGHT="2", GPUART_HIP_MAX_BATCH="1")
    assert p.returncode == 0 and "gathered 8K" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


def test_random_frames_ranks_and_roots(rccl_stub):
    """Twenty-four random decompositions in one process: N in 2..8 ranks, frames of 8..260 x 8..150 pixels (ragged widths, ragged last bands, more
    ranks than bands -> empty shares, the root's own share empty), a random root, one or two passes; every gathered frame == the single-context
    frame, bit for bit, and every non-empty peer sent exactly its rows."""
    p = _child(_SETUP + """
rng = np.random.RandomState(20261005)
done = 0
for case in range(24):
    N = int(rng.randint(2, 9)); W = int(rng.randint(8, 261)); H = int(rng.choice([8, 9, 15, 16, 17, 40, 63, 64, 65, 100, 150])); K = int(rng.randint(1, 3))
    root = int(rng.randint(N))
    one = renderer(W, H); render(one, K)
    want = one.read_radiance(True)
    ranks = [renderer(W, H, (k, N)) for k in range(N)]
    shares = [B.share_of_rank(W, H, k, N) for k in range(N)]
    for r in ranks: render(r, K)
    bes = [r.backend for r in ranks]
    B.comm_init_all(bes)
    before = served()
    full = B.gather_all_read(bes, 1, float(K), root, W, H)
    after = served()
    assert same(full, want), "case %d (N %d, %dx%d, root %d): %d pixels differ" % (case, N, W, H, root, int((full.view(np.uint32) != want.view(np.uint32)).any(-1).sum()))
    assert after[0] - before[0] == sum(1 for k, g in enumerate(shares) if k != root and g.th)
    assert after[1] - before[1] == 16 * W * sum(g.th for k, g in enumerate(shares) if k != root)
    for b in bes: b.comm_destroy()
    for r in ranks + [one]: r.close()
    done += 1
print("random gathers", done)
""", rccl_stub, timeout=600, GPUART_HIP_PASSES_IN_FLIGHT="2", GPUART_HIP_MAX_BATCH="2")
    assert p.returncode == 0 and "random gathers 24" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


@pytest.mark.parametrize("N", [2, 3, 8])
def test_one_thread_per_rank_gathers_the_same_frame(rccl_stub, N):
    """_comm_init + _gather: every rank on its own thread with its own context, the way bench.py's ranks and any one-process-per-GPU
    caller use the library — ncclCommInitRank joins the ranks, the shares and verdicts travel by ncclAllGather (N entries), peers send,
    the root receives at the offsets the share table gives and scatters into a device frame. 72 x 60 = 7.5 bands: ragged last band for
    every N, an empty share never (N = 8: rank 7 holds the 4-row band)."""
    p = _child(_SETUP + """
N, W, H, K = %d, 72, 60, 2
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
one = renderer(W, H)
uid = B.comm_unique_id()
ranks = [renderer(W, H, (k, N)) for k in range(N)]
for root in (0, N - 1):
    render(one, K)                             # (a Renderer draws fresh RandSeeds for every sequence: the ranks' second one is `one`'s second one)
    want = one.read_radiance(True)
    frame = C.c_void_p()
    assert hip.hipMalloc(C.byref(frame), W * H * 16) == 0 and hip.hipMemset(frame, 0xff, W * H * 16) == 0
    assert hip.hipDeviceSynchronize() == 0     # (the memset runs on the null stream, the library's streams do not wait for that one)
    errors = []
    def rank(k):
        try:
            b = ranks[k].backend
            if root == 0:
                b.comm_init(N, k, uid)
                assert b.comm_info() == (N, k)
            render(ranks[k], K)
            b.gather(1, float(K), root, frame.value if k == root else 0)
            b.wait(60000)
        except Exception as e:
            errors.append((k, repr(e)))
    before = served()
    ts = [threading.Thread(target=rank, args=(k,)) for k in range(N)]
    for t in ts: t.start()
    for t in ts: t.join(120)
    assert not errors and not any(t.is_alive() for t in ts), errors
    after = served()
    assert after[2] - before[2] == 1 and after[0] - before[0] == N - 1, (before, after)     # one all-gather of the shares, N-1 transfers
    full = np.empty((H, W, 4), np.float32)
    assert hip.hipMemcpy(full.ctypes.data_as(C.c_void_p), frame, W * H * 16, 2) == 0
    assert same(full, want), "root %%d: %%d pixels differ" %% (root, int((full.view(np.uint32) != want.view(np.uint32)).any(-1).sum()))
# a rank that cannot take part (the root without a frame buffer) makes EVERY rank fail, nobody is left in a receive
codes = {}
def bad(k):
    try:
        ranks[k].backend.gather(1, float(K), 0, 0)
        codes[k] = 0
    except B.HipError as e:
        codes[k] = str(e)
ts = [threading.Thread(target=bad, args=(k,)) for k in range(N)]
for t in ts: t.start()
for t in ts: t.join(120)
assert not any(t.is_alive() for t in ts)
assert "frame buffer" in codes[0] and all("could not prepare" in codes[k] for k in range(1, N)), codes
assert served()[0] == after[0]                                                              # and nothing was transferred
for r in ranks: r.backend.comm_destroy()
print("gathered per rank")
""" % N, rccl_stub)
    assert p.returncode == 0 and "gathered per rank" in p.stdout, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])


@pytest.mark.parametrize("N", [2, 3, 8])
def test_gpuart_cli_gpus_n_writes_the_single_gpu_frame(rccl_stub, tmp_path, N):
    """gpuart_cli --gpus N (Renderer::SetShare, GatherRadiance = ncclCommInitAll + gpuart_hip_gather_all_read, ReleaseCommunicator), every
    rank on device 0: the PFM equals the single-GPU PFM byte for byte. 200 x 136: 17 bands."""
    base = [CLI, "--scene", "box", "--width", "200", "--height", "136", "--mode", "pt", "--spp", "4", "--user-sphere", "-0.4,0,0.2,0.25,2"]
    one, many = str(tmp_path / "one.pfm"), str(tmp_path / "many.pfm")
    p = subprocess.run(base + ["--pfm", one], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-3000:]
    env = dict(os.environ, GPUART_HIP_RCCL_LIBRARY=rccl_stub, GPUART_HIP_TEST_SHARED_DEVICE="1", GPUART_CLI_SHARED_DEVICE="1")
    p = subprocess.run(base + ["--gpus", str(N), "--pfm", many], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    for ph in ("frame gather", "communicator init", "communicator destroy"):
        assert "gpuart phase begin: %s" % ph in p.stderr and "gpuart phase end:   %s" % ph in p.stderr, p.stderr[-3000:]
    assert '"gpus": %d' % N in p.stdout
    assert open(one, "rb").read() == open(many, "rb").read(), "%d ranks: the gathered frame differs from the single-GPU frame" % N
