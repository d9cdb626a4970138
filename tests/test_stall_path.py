"""The waits of the multi-GPU read-out are finite, and a stalled one says where.

The path gpuart_cli --gpus N takes — ncclCommInitAll -> gpuart_hip_gather_all_read -> ncclCommDestroy, the step the reference performs
with its ptracingNormalize draw (src/renderer.cpp:601-616) inside the loop of src/main.cpp:549-599 — had three waits without a bound,
and one child of round 4 did not end. Here every one of them is HELD and must come back within its bound:
  * without a GPU: the mechanism (a held call under gpuart_hip's bounded(), the phase watchdog's exit);
  * on the GPU (-m gpu, marked rccl: they run last): the real entry points against tests/stubs/rccl_stub.cpp — an RCCL stand-in whose
    ncclCommInitAll / ncclGroupEnd / ncclCommDestroy can be told never to return — and the read-back wait of gpuart_hip_gather_all_read
    against a stream held by gpuart_hip_test_stall; then gpuart_cli as a child process in each of those situations.
Nothing here tries to make the round-4 stall show again."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CLI = os.path.join(ROOT, "gpuart_amd", "bin", "gpuart_cli")


def _child(code, env=None, timeout=60):
    """A fresh python process (the watchdog ends the process it fires in)."""
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % ROOT + code], capture_output=True, text=True,
                       env=dict(os.environ, **(env or {})), timeout=timeout)
    return p, time.perf_counter() - t0


# ---- no GPU needed -----------------------------------------------------------------------------------------------------------
def test_a_held_call_comes_back_within_its_bound():
    from gpuart_amd import binding as B
    L = B.hip_lib()
    t0 = time.perf_counter()
    assert L.gpuart_hip_test_bounded_call(30, 2000, 0) == 0                      # returns in time: its own result
    assert time.perf_counter() - t0 < 1.0
    t0 = time.perf_counter()
    assert L.gpuart_hip_test_bounded_call(3000, 150, 0) == B.ERR_TIMEOUT          # held: the bound, not the call, ends the wait
    dt = time.perf_counter() - t0
    assert 0.14 < dt < 1.0, dt
    msg = L.gpuart_hip_last_error().decode()
    assert "has not returned after 150 ms" in msg and "gpuart_hip_test_bounded_call" in msg, msg
    assert not B.comm_stuck()
    t0 = time.perf_counter()
    assert L.gpuart_hip_test_bounded_call(40, 0, 0) == 0                          # bound 0: inline, unbounded
    assert 0.03 < time.perf_counter() - t0 < 1.0


def test_bounded_calls_of_several_threads_run_side_by_side():
    """Round 6: every calling thread has its own helper. Six threads each hold a bounded call for 300 ms: together they take ~300 ms, not
    1.8 s — the ranks of a one-thread-per-rank process are all inside ncclCommInitRank at once, each waiting for the others; behind ONE helper
    and one lock (round 5) the first rank was let in and the peers it waited for were kept outside (tests/test_gather_inprocess.py found it)."""
    import threading
    from gpuart_amd import binding as B
    L = B.hip_lib()
    rcs = []
    def call():
        rcs.append(L.gpuart_hip_test_bounded_call(300, 5000, 0))
    ts = [threading.Thread(target=call) for _ in range(6)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join(30)
    dt = time.perf_counter() - t0
    assert rcs == [0] * 6 and 0.29 < dt < 1.2, (rcs, dt)
    assert not B.comm_stuck()


def test_after_a_call_that_never_returned_the_communicator_layer_refuses_at_once():
    p, dt = _child("""
from gpuart_amd import binding as B
import time
L = B.hip_lib()
assert L.gpuart_hip_test_bounded_call(60000, 100, 1) == B.ERR_TIMEOUT
assert B.comm_stuck()
t0 = time.perf_counter()
import ctypes as C
ctx = C.c_void_p(1)                  # never dereferenced: the layer answers before it looks at the context
rc = L.gpuart_hip_comm_init(ctx, 1, 0, (C.c_ubyte * 128)())
assert rc == B.ERR_TIMEOUT, rc
assert "out of service" in L.gpuart_hip_last_error().decode()
assert time.perf_counter() - t0 < 1.0
print("refused")
""")
    assert p.returncode == 0 and "refused" in p.stdout, (p.stdout, p.stderr)


def test_the_watchdog_names_the_phase_and_the_recent_errors_and_ends_the_process():
    p, dt = _child("""
from gpuart_amd import binding as B
import time
L = B.hip_lib()
with B.phase("a phase that ends", 5000):
    time.sleep(0.05)
L.gpuart_hip_test_bounded_call(60000, 50, 1)       # leaves an error in the library's log and the layer stuck
B.phase_begin("the phase that hangs", 300)
time.sleep(20)
print("not reached")
""")
    assert p.returncode == 86, (p.returncode, p.stderr)   # GPUART_HIP_WATCHDOG_EXIT_CODE
    assert dt < 10, dt
    assert "not reached" not in p.stdout
    err = p.stderr
    assert "gpuart phase begin: a phase that ends" in err and "gpuart phase end:   a phase that ends" in err, err
    assert "gpuart phase begin: the phase that hangs" in err, err
    assert "gpuart watchdog" in err and "phase 'the phase that hangs'" in err, err
    assert "has not returned after 50 ms" in err and "stuck in: gpuart_hip_test_bounded_call" in err, err


def test_a_phase_that_ends_in_time_disarms_the_watchdog():
    p, dt = _child("""
from gpuart_amd import binding as B
import time
B.phase_begin("short", 200)
time.sleep(0.05)
B.phase_end()
time.sleep(0.6)
B.phase_begin("lines only", 0)
time.sleep(0.3)
print("alive")
""", env={"GPUART_HIP_PHASE_LOG": "0"})
    assert p.returncode == 0 and "alive" in p.stdout, (p.returncode, p.stderr)
    assert "gpuart phase" not in p.stderr  # GPUART_HIP_PHASE_LOG=0


# ---- the real entry points, on the GPU, against the stand-in RCCL ----------------------------------------------------------------
@pytest.fixture(scope="module")
def stub(rccl_stub):
    return rccl_stub


_SETUP = """
import time, numpy as np
from gpuart_amd import binding as B
from gpuart_amd import synth_scenes as S
W, H = 72, 40
cam = dict(S.DEFAULT_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam, device=0)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.init_box()
r.restart_path_tracing(1, 2)
r.path_tracing_pass(); r.path_tracing_pass()
own = r.read_radiance(True)
be = r.backend
"""


def _held(what, stub, code, timeout_env, expect):
    env = {"GPUART_HIP_RCCL_LIBRARY": stub, "RCCL_STUB_HOLD": what}
    env.update(timeout_env)
    p, dt = _child(_SETUP + code, env=env, timeout=120)
    assert expect in p.stdout, (what, p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return p, dt


@pytest.mark.gpu
@pytest.mark.rccl
def test_with_nothing_held_the_stand_in_serves_the_read_out(stub):
    """Control: the same child, nothing held — the frame gathered through ncclCommInitAll + gpuart_hip_gather_all_read (one rank) is the
    renderer's own read-back, and the communicator is destroyed. (The stand-in really is what the library loaded.)"""
    _held("none", stub, """
assert B.comm_library().endswith("librccl_stub.so"), B.comm_library()
B.comm_init_all([be])
full = B.gather_all_read([be], 1, 2.0, 0, W, H)
assert (full.view(np.uint32) == own.view(np.uint32)).all()
be.comm_destroy()
assert not B.comm_stuck()
print("served")
""", {"GPUART_HIP_COMM_TIMEOUT_MS": "5000"}, "served")


@pytest.mark.gpu
@pytest.mark.rccl
@pytest.mark.parametrize("what,call", [("init_all", "ncclCommInitAll"), ("group_end", "ncclGroupStart .. ncclGroupEnd"), ("destroy", "ncclCommDestroy")])
def test_a_held_rccl_call_returns_a_timeout_that_names_it(stub, what, call):
    """ncclCommInitAll, the transfers' ncclGroupEnd and ncclCommDestroy, each held for an hour by the stand-in: the library call that
    wraps it returns GPUART_HIP_ERR_TIMEOUT within GPUART_HIP_COMM_TIMEOUT_MS, names the RCCL call, and the layer refuses whatever
    comes next; the context still renders and reads back (single-GPU work does not touch the layer)."""
    p, dt = _held(what, stub, """
t0 = time.perf_counter()
err = None
try:
    B.comm_init_all([be])
    full = B.gather_all_read([be], 1, 2.0, 0, W, H)
    be.comm_destroy()
except B.HipError as e:
    err = e
dt = time.perf_counter() - t0
assert err is not None and err.code == B.ERR_TIMEOUT, err
assert %r in str(err) and "has not returned after 400 ms" in str(err), str(err)
assert 0.39 < dt < 5.0, dt
assert B.comm_stuck()
try:
    B.comm_init_all([be])
    raise SystemExit("the stuck layer accepted a new communicator")
except B.HipError as e:
    assert e.code == B.ERR_TIMEOUT and "out of service" in str(e), str(e)
again = r.read_radiance(True)
assert (again.view(np.uint32) == own.view(np.uint32)).all()
print("bounded in %%.2f s" %% dt)
import os; os._exit(0)     # (a parked helper thread sits in the stand-in: no interpreter tear-down around it)
""" % call, {"GPUART_HIP_COMM_TIMEOUT_MS": "400"}, "bounded in")
    assert dt < 60, dt


@pytest.mark.gpu
@pytest.mark.rccl
def test_the_read_back_wait_of_gather_all_read_is_bounded(stub):
    """The wait that was a plain hipStreamSynchronize (gpuart_hip.hip:1330 of round 4): the root's stream is held — as rows that never
    arrive would hold it — and gpuart_hip_gather_all_read comes back with GPUART_HIP_ERR_TIMEOUT after GPUART_HIP_GATHER_TIMEOUT_MS;
    when the stream frees itself the same call works, and comm_destroy of the context that gave up is bounded too."""
    _held("none", stub, """
B.comm_init_all([be])
full = B.gather_all_read([be], 1, 2.0, 0, W, H)     # (first use allocates the send / staging buffers: hipMalloc and hipFree may wait for the device)
assert (full.view(np.uint32) == own.view(np.uint32)).all()
be.test_stall(1500)
t0 = time.perf_counter()
try:
    B.gather_all_read([be], 1, 2.0, 0, W, H)
    raise SystemExit("no timeout")
except B.HipError as e:
    dt = time.perf_counter() - t0
    assert e.code == B.ERR_TIMEOUT and "not complete after 200 ms" in str(e), str(e)
    assert 0.19 < dt < 1.2, dt
t0 = time.perf_counter()
try:
    be.comm_destroy()            # the stream is still held: bounded, not a hang
    raise SystemExit("comm_destroy did not notice the abandoned gather")
except B.HipError as e:
    assert e.code == B.ERR_TIMEOUT, str(e)
    assert time.perf_counter() - t0 < 1.2
be.wait(10000)                   # the stall ends by itself
full = B.gather_all_read([be], 1, 2.0, 0, W, H)
assert (full.view(np.uint32) == own.view(np.uint32)).all()
be.comm_destroy()
print("bounded")
""", {"GPUART_HIP_GATHER_TIMEOUT_MS": "200", "GPUART_HIP_COMM_TIMEOUT_MS": "5000"}, "bounded")


def _cli(stub, hold, env, timeout=120):
    e = dict(os.environ, GPUART_HIP_RCCL_LIBRARY=stub, RCCL_STUB_HOLD=hold, GPUART_CLI_FORCE_GATHER="1")
    e.update(env)
    t0 = time.perf_counter()
    p = subprocess.run([CLI, "--scene", "box", "--width", "72", "--height", "40", "--mode", "pt", "--spp", "3"], capture_output=True, text=True, env=e, timeout=timeout)
    return p, time.perf_counter() - t0


@pytest.mark.gpu
@pytest.mark.rccl
@pytest.mark.parametrize("hold,phase,call", [
    ("init_all", "communicator init", "ncclCommInitAll"),
    ("group_end", "frame gather", "ncclGroupStart .. ncclGroupEnd"),
    ("destroy", "communicator destroy", "ncclCommDestroy")])
def test_gpuart_cli_ends_within_its_bounds_and_says_where(stub, hold, phase, call):
    """The child process of round 4's recorded stall, with each of its three waits held in turn: it prints the phase it entered, the
    library's bounded call gives up after GPUART_HIP_COMM_TIMEOUT_MS naming the RCCL call, and the process ends non-zero without
    unwinding — seconds, not the 180 s after which the test harness used to kill it."""
    p, dt = _cli(stub, hold, {"GPUART_HIP_COMM_TIMEOUT_MS": "500"})
    assert p.returncode == 1, (p.returncode, p.stderr[-3000:])
    assert dt < 60, dt
    assert "gpuart phase begin: %s" % phase in p.stderr, p.stderr[-3000:]
    assert call in p.stderr and "has not returned after 500 ms" in p.stderr, p.stderr[-3000:]
    assert "ending without unwinding" in p.stderr, p.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.rccl
def test_gpuart_cli_watchdog_ends_a_phase_nothing_else_bounds(stub):
    """Belt and braces: with the library's own bound switched off (GPUART_HIP_COMM_TIMEOUT_MS=0: the RCCL call runs inline, as it did in
    round 4) the held ncclCommInitAll blocks the main thread — the watchdog thread sees the phase outlive GPUART_PHASE_TIMEOUT_MS,
    prints it and ends the process with its own exit code."""
    p, dt = _cli(stub, "init_all", {"GPUART_HIP_COMM_TIMEOUT_MS": "0", "GPUART_PHASE_TIMEOUT_MS": "700"})
    assert p.returncode == 86, (p.returncode, p.stderr[-3000:])
    assert dt < 60, dt
    assert "gpuart watchdog" in p.stderr and "phase 'communicator init" in p.stderr, p.stderr[-3000:]


@pytest.mark.gpu
@pytest.mark.rccl
def test_gpuart_cli_through_the_stand_in_with_nothing_held(stub, tmp_path):
    """Control for the three above: same child, nothing held — exit 0, every phase begins and ends, same PFM as the plain read-back."""
    a, b = str(tmp_path / "a.pfm"), str(tmp_path / "b.pfm")
    base = [CLI, "--scene", "box", "--width", "72", "--height", "40", "--mode", "pt", "--spp", "3"]
    p = subprocess.run(base + ["--pfm", a], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    p = subprocess.run(base + ["--pfm", b], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, GPUART_HIP_RCCL_LIBRARY=stub, GPUART_CLI_FORCE_GATHER="1"))
    assert p.returncode == 0, p.stderr
    for ph in ("frame gather", "communicator init", "communicator destroy"):
        assert "gpuart phase begin: %s" % ph in p.stderr and "gpuart phase end:   %s" % ph in p.stderr, p.stderr
    assert open(a, "rb").read() == open(b, "rb").read()
