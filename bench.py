#!/usr/bin/env python3
"""bench.py — the hot path measured as BASELINE.json asks: Mrays/s (+ ms/frame) of progressive path tracing, dragon-class
scene (Scene D: 100 352-triangle mesh + floor disc), 1920x1080, path depth 8, 1 path per pixel per pass, on N GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
        N > 1: one process per GPU. Started by `python -m torch.distributed.run ... bench.py --gpus N ...` the ranks are
        taken from the environment; started plainly, bench.py starts that launcher itself (before anything touches a GPU)
        and relays rank 0's line — it never reports n_gpus != the ranks that rendered.

A "step" is one progressive pass (gpuart::Renderer::RenderPathTracingPass) over the whole frame. With N > 1 the FIXED frame
is sharded by rows (8-row bands dealt round-robin, gpuart_hip_share_of_rank; strong scaling, north_star's "tile scaling");
after the K timed passes every rank hands its normalised rows to the library's RCCL gather (gpuart_hip_gather, inside the
timed region).

What is counted:
  * `value` = rays EXECUTED / wall time: the BVH queries (camera, bounce and Sun-shadow rays) the timed fast mode really
    performs for these K passes, counted exactly by the device in an untimed replay (mode 4). The reference itself performs
    more — the fast mode renders bit-identical images but skips Sun-shadow queries that cannot matter and stops them at the
    first hit —: the reference-defined count (untimed mode-1 replay) over the same wall time is reported beside
    it as `mrays_reference_defined_per_s`, never as the headline.
  * The K-pass timed sequence is run `--repeats` times (default 5), each bracketed by barrier + synchronize: `ms_per_step` is
    the median repetition, `ms_per_step_spread` the fastest and slowest one.
  * `roofline`: device-level fractions, counters collected by rocprofv3 child runs of THIS invocation on the same passes and
    divided by ms_per_step (overlapping launches are never double counted). The traversal step of the BVH queries is balanced
    on two units: `frac` = VALU issue (SQ_INSTS_VALU against the guide's 2 cycles per wave64
    instruction; also against the rates tools/ubench measures on the box), `l1_accesses` = vector-L1 cache accesses against the
    highest rate measured for the product's own node fetch, `node_visits` = executed node visits against the rate of the same
    step in a register-resident micro-benchmark. The HBM view (`hbm`, `traffic`) is secondary: the 8.7 MB tree is
    cache-resident, so algorithmic bytes / time exceeds the HBM peak and is not a fraction of anything.
Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

# The renderer keeps up to 8 pipeline runs in flight on 9 HIP streams; with ROCm's default of 4 hardware queues their kernels would
# serialise. Must be set before the HIP runtime initialises (i.e. before torch is imported). See DESIGN.md section 4.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from gpuart_amd import bench_line as BL  # the roofline block's definitions and peaks (pure functions: tests/test_bench_line.py drives them)

# The reference's own GLSL on Mesa llvmpipe, measured in the BUILD CONTAINER (8 vCPU; tests/golden/time_llvmpipe.py, output in
# profiles/r02/llvmpipe_reference_glsl_container.txt). /root/reference cannot travel to the GPU box, so this is a recorded figure,
# labelled as such; `cpu_baseline` is the figure measured in this run on this box.
LLVMPIPE_CONTAINER = {
    "cfg3": {"value": 0.957, "ms_per_frame": 5428.8},
    "cfg2": {"value": 3.315, "ms_per_frame": 1409.3},
    "dragon871k": {"value": 0.737, "ms_per_frame": 7073.4},  # profiles/r03/llvmpipe_reference_glsl_container_871k.txt
}

WORKLOADS = {
    "cfg3": dict(text="Scene D (dragon-class, 100352 triangles + floor disc)", segs=8, camera="benchmark camera"),
    "cfg2": dict(text="Scene P (256 spheres + 16 discs)", segs=4, camera="default camera"),
    "dragon871k": dict(text="the reference's largest scene at its size (src/main.cpp:321 \"dragon 871k\"): 871200-triangle stand-in + floor disc, "
                            "75 MB of tree on the device (it leaves the L2s)", segs=8, camera="benchmark camera"),
    "cluster": dict(text="InitCluster on the synthetic cluster_100k stand-in (100000 spheres + floor disc)", segs=5, camera="near camera"),
    "tree": dict(text="InitTree on the synthetic tree1_21k stand-in (9841 cones + 9775 spheres + floor disc)", segs=5, camera="near camera"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--frame", default="1920x1080",
                    help="frame size WxH (default: cfg3's 1080p, the configuration the metric is quoted on; 3840x2160 = "
                         "the 4K frame north_star also asks for — see DESIGN.md for its numbers)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg3",
                    help="cfg3 (default, the configuration the metric is quoted on): Scene D, depth 8; cfg2 of BASELINE.json: Scene P, "
                         "depth 4; cluster / tree: the reference's two primitive-list scenes (InitCluster / InitTree) on seeded stand-ins; "
                         "dragon871k: cfg3's workload on a mesh of the real Stanford dragon's size (the tree leaves the L2s)")
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-pass timed sequence is run (median reported, min/max beside it)")
    ap.add_argument("--no-verify-gather", action="store_true", help="N > 1: skip the untimed check of the gathered frame against rank 0's own whole-frame render")
    ap.add_argument("--gather-timeout", type=float, default=120.0,
                    help="N > 1: seconds a rank waits for the frame gather before it names the ranks that have not arrived and exits non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 child runs (roofline counters become null)")
    ap.add_argument("--verify-gather", action="store_true", help="(kept for old command lines: the check is always made for N > 1 now)")
    ap.add_argument("--render-only", action="store_true",
                    help="(internal) render warm-up + steps passes of the workload and exit: the program the rocprofv3 child runs profile")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves (before any GPU call)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd).returncode)


def make_renderer(args, W, H, device, tmpdir):
    """The workload's Renderer (gpuart::Renderer API; scene build is not part of the timed region) + what the oracle needs."""
    from gpuart_amd import binding as B
    from gpuart_amd import synth_scenes as S
    w = args.workload
    cam = dict({"cfg3": S.BENCH_CAMERA, "dragon871k": S.BENCH_CAMERA, "cfg2": S.DEFAULT_CAMERA, "cluster": S.CLUSTER_NEAR_CAMERA, "tree": S.TREE_NEAR_CAMERA}[w])
    cam["dir"] = S.camera_dir(cam)
    t0 = time.time()
    r = B.Renderer(W, H, cam, device=device)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    if w in ("cfg3", "cfg2", "dragon871k"):
        descs = S.scene_d() if w == "cfg3" else S.scene_d(660, 660) if w == "dragon871k" else S.scene_p()
        r.set_primitives(B.make_prims(descs))
    else:  # through the reference's loader path: a .dat file, InitCluster / InitTree (src/scenes.cpp:69-103)
        lines = S.cluster_dat_lines() if w == "cluster" else S.tree_dat_lines()
        path = os.path.join(tmpdir, w + ".dat")
        S.write_lines(path, lines)
        assert (r.init_cluster(path) if w == "cluster" else r.init_tree(path)), "scene file did not load"
        descs = S.dat_descs(lines, **(S.CLUSTER_LOAD if w == "cluster" else S.TREE_LOAD)) + [S.FLOOR_DISC_CT]
    r.set_max_path_segments(WORKLOADS[w]["segs"])
    assert r.is_ok()
    return r, cam, descs, time.time() - t0


def run_passes(r, n):
    r.restart_path_tracing(1, n)
    for _ in range(n):
        r.path_tracing_pass()


PROFILE_REPEATS = 3  # the profiled children render the timed shape — K passes between two observations — this many times


def profile_children(args, K):
    """rocprofv3 child runs of `bench.py --render-only`: every child renders EXACTLY the timed shape (a sequence of K passes, PROFILE_REPEATS
    times) and nothing else, so counters / (K x PROFILE_REPEATS) describe the schedule that ms_per_step times. One --pmc pass per
    counter set (BL.PMC_SETS, filtered by what `--list-avail` offers: FETCH_SIZE and WRITE_SIZE do not share TCC slots, MI355X_MICROARCH.md
    "rocprofv3 PMC slots") + one un-instrumented --kernel-trace pass for per-kernel durations in the real, overlapping schedule.
    Returns (prof, trace, passes, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, 0, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--render-only", "--steps", str(K), "--repeats", str(PROFILE_REPEATS), "--workload", args.workload,
             "--frame", args.frame]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        avail = subprocess.run([exe, "--list-avail"], cwd="/tmp", env=env, capture_output=True, text=True, timeout=120).stdout
    except Exception:  # noqa: BLE001
        avail = ""
    prof, trace = {}, None
    for tag, counters in BL.PMC_SETS + (("trace", None),):
        if counters is not None:
            counters = BL.pick_available(counters, avail)
            if not counters:
                continue
        d = tempfile.mkdtemp(prefix="gpuart_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace", "--output-format", "csv"] + (["--pmc"] + counters if counters else []) + ["-d", d, "-o", "x", "--"] + child
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                return None, None, 0, "rocprofv3 %s failed (rc %d): %s" % (tag, p.returncode, (p.stderr or p.stdout)[-300:])
            if counters:
                prof[tag] = BL.read_counters(d)
            else:
                trace = BL.read_kernel_trace(d)
        except Exception as e:  # noqa: BLE001
            return None, None, 0, "rocprofv3 child run failed: %r" % (e,)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return prof, trace, K * PROFILE_REPEATS, ("rocprofv3 child runs of this invocation (`%s`): %d sequences of %d passes each, the timed shape; sums over all "
                                               "dispatches / %d passes" % (" ".join(child[1:]), PROFILE_REPEATS, K, K * PROFILE_REPEATS))


def main():
    args = parse_args()
    W, H = (int(x) for x in args.frame.lower().split("x"))
    K, Wm = args.steps, args.warmup
    segs = WORKLOADS[args.workload]["segs"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)  # does not return

    from gpuart_amd import binding as B
    from gpuart_amd import sharding
    from gpuart_amd import synth_scenes as S
    tmpdir = tempfile.mkdtemp(prefix="gpuart_bench_")

    if args.render_only:  # the profiled child: same scene, same seeds, same pass sequence; nothing else
        r, _, _, _ = make_renderer(args, W, H, 0, tmpdir)
        for _ in range(max(1, args.repeats)):
            r.set_seed(5489)
            run_passes(r, K)
            r.finish()
        r.close()
        shutil.rmtree(tmpdir, ignore_errors=True)
        return

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: the line would not describe the ranks that rendered" % (args.gpus, world))
    dist = None
    # GPUART_BENCH_BACKEND=gloo rehearses the N>1 code path on a box with fewer GPUs than ranks (ranks share GPUs, the exchange is
    # staged through host memory by sharding.gather_shares_host); the driver's runs use RCCL ("nccl").
    backend = os.environ.get("GPUART_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    # Every step of a rank that waits for other ranks is a named phase (gpuart_hip_phase_begin: one line on stderr when it begins and
    # ends, and the library's watchdog thread behind it — a phase that outlives its bound prints the phase and the library's recent
    # errors and ends THIS rank with exit code 86; nothing is re-executed in a process that has touched the GPU). N = 1 has no such step.
    phase_ms = int(float(os.environ.get("GPUART_BENCH_PHASE_TIMEOUT_S", "300")) * 1000)

    def phase(name, ms=None):
        return B.phase("rank %d: %s" % (rank, name), phase_ms if ms is None else ms)

    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        with phase("torch.distributed init (%s)" % backend):
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")  # where tensors exchanged through torch.distributed live
    torch.cuda.set_device(dev)

    r, cam, scene_descs, setup_s = make_renderer(args, W, H, local_rank, tmpdir)
    be = r.backend
    info = be.scene_info()

    # ---- this rank's share: 8-row bands of the frame dealt round-robin to the ranks (balances sky / floor / mesh rows
    #      statistically; 8x8 pixel tiles stay intact); nothing is exchanged until the final gather ----
    share = B.share_of_rank(W, H, rank, world)
    th = share.th
    gather_note = None
    if world > 1:
        assert th > 0, "frame too small for %d ranks" % world
        assert r.set_interleaved_tile(share.x0, share.y0, share.tw, share.th, share.band_rows, share.band_stride)
        if backend == "nccl":
            # the library's own communicator: rank 0's id travels through torch.distributed, then ncclCommInitRank per rank
            idt = torch.zeros(128, dtype=torch.uint8, device=xdev)
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(B.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            try:
                with phase("communicator init (gpuart_hip_comm_init = ncclCommInitRank)"):
                    be.comm_init(world, rank, bytes(idt.cpu().numpy().tobytes()))
            except B.HipError as e:
                if e.code == B.ERR_TIMEOUT:  # ncclCommInitRank never returned (a rank is missing): its thread is parked, nothing to fall back to
                    print("bench.py rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
                    os._exit(3)
                gather_note = "gpuart_hip_comm_init failed (%s): gathered through torch.distributed point-to-point instead" % e
            # all ranks must take the same exchange path: one rank's failure moves every rank to the fallback
            ok = torch.tensor([1 if gather_note is None else 0], dtype=torch.int32, device=xdev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 0 and gather_note is None:
                be.comm_destroy()
                gather_note = "gpuart_hip_comm_init failed on another rank: gathered through torch.distributed point-to-point instead"
        else:
            gather_note = "GPUART_BENCH_BACKEND=%s rehearsal: gathered through host memory" % backend

    # ---- exact work counts: the same K passes, untimed; reference-defined (mode 1) and as the fast mode executes them (mode 4) ----
    replay_frames = {}

    def count(mode):
        r.set_seed(5489)
        be.set_mode(mode)
        be.counters(reset=True)
        run_passes(r, K)
        be.finish()
        c = be.counters(reset=True)
        replay_frames[mode] = r.read_radiance(False)  # this rank's accumulator after the K passes (frames_verified below)
        be.set_mode(0)
        return [c.rays, c.nodes, c.prim_tests[0], c.prim_tests[1], c.prim_tests[2], c.prim_tests[3], c.segments,
                c.algorithmic_bytes() + 32 * W * th * K, c.rewalks, c.box_steps]

    counts = torch.tensor(count(1) + count(4), dtype=torch.float64, device=xdev)
    if dist is not None:
        dist.all_reduce(counts)
    ref, exe = counts[:10].tolist(), counts[10:].tolist()
    rays, nodes, segments, alg_bytes = ref[0], ref[1], ref[6], ref[7]

    # ---- warm-up (untimed), then EXACTLY K timed passes ----
    full = torch.empty((H, W, 4), dtype=torch.float32, device=dev) if (world > 1 and rank == 0) else None
    host_tile = None

    store = None
    if dist is not None:
        try:
            store = dist.distributed_c10d._get_default_store()  # TCP: independent of RCCL, so it still answers when a collective hangs
        except Exception:  # noqa: BLE001
            store = None
    gather_seq = [0]
    # a rank's arrival at gather #n is recorded in the rendezvous store by a helper thread: the TCP round trip of store.set (~0.1 ms) stays
    # out of the timed region, where a rank's whole job is ~3 ms at N = 8
    import queue
    import threading
    arrivals = queue.Queue()

    def _announce():
        while True:
            key = arrivals.get()
            try:
                store.set(key, "1")
            except Exception:  # noqa: BLE001
                pass
    if store is not None:
        threading.Thread(target=_announce, daemon=True).start()

    def missing_ranks(tag):
        if store is None:
            return "unknown (no rendezvous store)"
        miss = []
        for k in range(world):
            try:
                if not store.check(["gpuart_arrived_%s_%d" % (tag, k)]):
                    miss.append(k)
            except Exception:  # noqa: BLE001
                miss.append(k)
        return miss

    def gather(divide_by):
        """Every rank's normalised rows -> rank 0's `full` (device). The library's RCCL gather; the announced fallback otherwise.
        Bounded: a rank that waits longer than --gather-timeout for its peers says which ranks never arrived and exits 3
        (nothing is restarted in a process that has touched the GPU)."""
        nonlocal host_tile
        gather_seq[0] += 1
        tag = str(gather_seq[0])
        if store is not None:
            arrivals.put("gpuart_arrived_%s_%d" % (tag, rank))
        if gather_note is None:
            try:
                # (the phase's bound sits ABOVE the library's own: its ERR_TIMEOUT comes first and names the missing ranks; the watchdog's
                #  exit is for what that does not cover — GPUART_BENCH_PHASE_TIMEOUT_S=0 must not make the two race)
                with phase("frame gather #%s (gpuart_hip_gather + gpuart_hip_wait)" % tag, int(args.gather_timeout * 1000) + max(phase_ms, 10000)):
                    be.gather(1, divide_by, 0, full.data_ptr() if rank == 0 else 0)
                    be.wait(int(args.gather_timeout * 1000))
            except B.HipError as e:
                if e.code == B.ERR_TIMEOUT:
                    print("bench.py rank %d: the frame gather did not complete within %.0f s (%s); ranks that never reached gather #%s: %s"
                          % (rank, args.gather_timeout, e, tag, missing_ranks(tag)), file=sys.stderr, flush=True)
                    os._exit(3)
                # any other failure: every rank says what ITS library call reported (the peers' calls fail with "rank k could not
                # prepare its share": the cause is in rank k's own message), then the rank ends non-zero
                print("bench.py rank %d: gpuart_hip_gather #%s failed: %s" % (rank, tag, e), file=sys.stderr, flush=True)
                raise
            return
        if host_tile is None:
            host_tile = torch.empty((th, W, 4), dtype=torch.float32, device=dev)
        be.export(1, host_tile.data_ptr(), divide_by)
        be.finish()
        fh = torch.empty((H, W, 4), dtype=torch.float32) if rank == 0 else None
        sharding.gather_shares_host(_P2P(dist, xdev), host_tile.cpu(), rank, world, W, H, fh)
        if rank == 0:
            full.copy_(fh)

    # The first pass sequence of a given shape uses every pass lane for the first time (its stream, its path buffers, the kernel
    # variants of this scene): ~3 ms once, which a W-pass warm-up — it runs on two lanes — does not take off the K timed passes.
    # One untimed sequence of the timed shape first, then the W warm-up passes the caller asked for.
    r.set_seed(5489)
    run_passes(r, K)
    be.finish()
    r.set_seed(5489)
    run_passes(r, Wm)
    be.finish()
    if dist is not None:
        # warm the exchange path too (the first transfer between two ranks sets up their channel) — and make sure it works on every
        # rank before anything is timed: an error of the library's gather on any rank moves ALL ranks to the announced fallback
        # (a timeout cannot be recovered from: that rank has already named the missing ranks and exited)
        failed = None
        try:
            gather(float(max(1, Wm)))
        except B.HipError as e:
            failed = str(e)
        ok = torch.tensor([0 if failed else 1], dtype=torch.int32, device=xdev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 0 and gather_note is None:
            gather_note = "gpuart_hip_gather failed in the warm-up (%s): gathered through torch.distributed point-to-point instead" % (failed or "on another rank")
            try:
                be.comm_destroy()
            except B.HipError:
                pass
            gather(float(max(1, Wm)))
    be.set_timing(2)
    be.kernel_time(0, reset=True)
    be.kernel_time(1, reset=True)
    reps = []  # per repetition: (max over ranks of the wall time, this rank's render time, this rank's gather time)
    phase_lines = B.phase_log(False)  # the timed gathers keep their watchdog and lose their two stderr lines (a rank's job is ~3 ms at N = 8)
    for _ in range(max(1, args.repeats)):
        r.set_seed(5489)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_passes(r, K)
        t_render = t_gather = None
        if dist is not None:
            be.finish()  # the rank's own passes (the gather would wait for them anyway)
            t_render = time.perf_counter() - t0
            gather(float(K))
            t_gather = time.perf_counter() - t0 - t_render
        be.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax[0])
        reps.append((dt, t_render, t_gather))
    B.phase_log(phase_lines)
    pass_ms, passes = be.kernel_time(0, reset=True)
    kernel_ms, launches = be.kernel_time(1, reset=True)
    be.set_timing(1)
    # ---- the frames that were just timed, checked (untimed): the accumulator the LAST timed repetition left on this rank against the
    #      accumulator of the mode-1 replay of the same K passes — reference-order walks, every reference query performed, the kernels
    #      the oracle is held against in tests/ — bit for bit; the mode-4 replay (whose ray count is `value`'s numerator) likewise. A
    #      line whose timed kernels rendered something else is not a measurement: the run exits non-zero after printing it.
    timed_frame = r.read_radiance(False)
    bad = {m: int((timed_frame[..., :3].view(np.uint32) != f[..., :3].view(np.uint32)).any(-1).sum()) for m, f in replay_frames.items()}
    rows_crc = int(np.bitwise_xor.reduce(timed_frame[..., :3].view(np.uint32).reshape(-1)))
    frames_bad = torch.tensor([bad.get(1, -1), bad.get(4, -1)], dtype=torch.float64, device=xdev)
    if dist is not None:
        dist.all_reduce(frames_bad)
    frames_bad = [int(x) for x in frames_bad.tolist()]
    frames_verified = frames_bad == [0, 0]
    del replay_frames, timed_frame
    order = sorted(range(len(reps)), key=lambda k: reps[k][0])
    med = order[len(order) // 2]
    elapsed = reps[med][0]
    elapsed_all = sum(x[0] for x in reps)
    spread = [round(reps[order[0]][0] / K * 1e3, 4), round(reps[order[-1]][0] / K * 1e3, 4)]
    multi = None
    if dist is not None:
        mine = torch.tensor([reps[med][1] * 1e3, reps[med][2] * 1e3], dtype=torch.float64, device=xdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        multi = {"per_rank_ms": [round(float(x[0]), 3) for x in every], "gather_ms": [round(float(x[1]), 3) for x in every],
                 "note": "the median repetition: per_rank_ms = each rank's own K passes of its share (submit to finish), gather_ms = from there "
                         "until gpuart_hip_gather had completed on that rank (rank 0: all rows received and placed; it includes waiting for the slowest rank)",
                 "rccl_ranks": None}
        if gather_note is None:
            multi["rccl_ranks"] = be.comm_info()[0]  # ncclCommCount of the library's communicator
            # which RCCL served the library's gather, and every librccl this process has mapped (torch.distributed brings its own
            # copy: two different files here would mean two RCCL instances sharing the GPU's queues)
            try:
                multi["rccl_library"] = B.comm_library()
            except B.HipError as e:
                multi["rccl_library"] = "unknown (%s)" % e
            try:
                multi["rccl_mapped"] = sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})
            except OSError:
                multi["rccl_mapped"] = None
        multi["gpu_max_hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")

    def comm_destroy():
        """gpuart_hip_comm_destroy (= ncclCommDestroy) as a watched phase; a destroy that does not return ends the rank (exit 3)."""
        try:
            with phase("communicator destroy (gpuart_hip_comm_destroy = ncclCommDestroy)"):
                be.comm_destroy()
        except B.HipError as e:
            print("bench.py rank %d: %s" % (rank, e), file=sys.stderr, flush=True)
            if e.code == B.ERR_TIMEOUT:
                os._exit(3)
            raise

    verify = dist is not None and not args.no_verify_gather
    if dist is not None and gather_note is None and not verify:
        comm_destroy()  # every rank, while all of them are still alive
    if rank != 0:
        if dist is not None:
            if verify:
                dist.barrier()  # rank 0 is still checking the gathered frame
                if gather_note is None:
                    comm_destroy()
            dist.destroy_process_group()
        return

    gather_verified = None
    if verify:
        # Once, untimed: the frame the last timed repetition gathered on rank 0 against rank 0's own render of the WHOLE frame (same
        # passes, same seeds): a gather that moved rows wrongly, or a rank that rendered something else, shows here instead of never.
        assert r.set_tile(0, 0, W, H)
        r.set_seed(5489)
        run_passes(r, K)
        whole = r.read_radiance(True)
        got = full.cpu().numpy()
        diff = (got[..., :3].view(np.uint32) != whole[..., :3].view(np.uint32)).any(-1)
        gather_verified = not bool(diff.any())
        multi["gather_verified"] = gather_verified
        multi["gather_verified_note"] = ("gathered frame == rank 0's own whole-frame render of the same passes, bit for bit" if gather_verified else
                                         "%d pixels differ; first differing frame rows: %s" % (int(diff.sum()), np.nonzero(diff.any(1))[0][:8].tolist()))
        print("verify-gather: gathered %d interleaved row sets == single-rank frame: %s" % (world, gather_verified), file=sys.stderr)
        dist.barrier()
        if gather_note is None:
            comm_destroy()

    # ---- one pass alone, observed after it (the reference's interactive loop, src/main.cpp:549-599): frame time, not throughput ----
    single_ms = None
    if world == 1:
        ts = []
        for _ in range(7):
            t1 = time.perf_counter()
            run_passes(r, 1)
            be.finish()
            ts.append((time.perf_counter() - t1) * 1e3)
        single_ms = float(np.median(ts[2:]))

    # ---- the opt-in nearest-child-first walk beside it (untimed for `value`): what exactness costs. The default walks every tree in the
    #      reference's order — the only order proven to return the reference's winner (csrc/hip/device_scene.h, tests/golden/order_adversary.npz)
    nearest = None
    if world == 1:
        be.set_nearest_first(1024)
        ts = []
        for _ in range(max(3, len(reps))):
            r.set_seed(5489)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_passes(r, K)
            be.finish()
            ts.append((time.perf_counter() - t1) / K * 1e3)
        nf_frame = r.read_radiance(False)
        be.set_nearest_first(0xffffffff)
        r.set_seed(5489)
        run_passes(r, K)
        same = not bool((nf_frame[..., :3].view(np.uint32) != r.read_radiance(False)[..., :3].view(np.uint32)).any())
        del nf_frame
        nearest = {"ms_per_step": round(float(np.median(ts)), 4), "same_frame_as_default": same,
                   "note": "gpuart_hip_set_nearest_first(1024): the round-4 default (nearer child first + certificate, ~17 %% fewer node visits), opt-in "
                           "since round 5 because the reference's phantom hits at grazing angles make any pruning walk in another order unprovable; "
                           "the same K passes, median of %d sequences; `same_frame_as_default`: its accumulator == the default walk's, bit for bit" % max(3, len(reps))}

    # ---- roofline (gpuart_amd/bench_line.py): counters from rocprofv3 child runs of this invocation that render the timed shape ----
    ms_step = elapsed / K * 1e3
    prof = trace = None
    passes_profiled = 0
    if world == 1 and not args.no_profile:
        prof, trace, passes_profiled, source = profile_children(args, K)
    else:
        source = "not collected (%s)" % ("--no-profile" if args.no_profile else "N > 1: one GPU's counters would not describe the job")
    roof = BL.assemble_roofline(ms_step, passes_profiled, prof, trace,
                                executed={"nodes": exe[1] / K, "steps": exe[9] / K, "algorithmic_bytes": exe[7] / K},
                                reference={"nodes": nodes / K, "algorithmic_bytes": alg_bytes / K},
                                kernel_events=(kernel_ms, launches, elapsed_all), passes_timed=K * len(reps))
    roof["source"] = source
    roof["kernel_ms_summed_per_pass"] = round(kernel_ms / (K * len(reps)), 4)  # HIP events, all repetitions

    # ---- CPU baseline: the oracle (port) on this box's host cores, bounded sample, rank 0, N=1 only ----
    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        cores = min(len(os.sched_getaffinity(0)), 16)  # the GPU box grants a 16-core CPU share per GPU
        otree, _ = O.build_bvh(scene_descs)
        c = O.camera(cam["pos"], cam["dir"], cam["up"], cam["fov_y"], cam["screen_dist"], W, H)
        sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
        P = O.make_params(sun, S.SUN_ALTITUDE, True, S.USER_SPHERE, 0.0, 0, float(c[12]), c[0:3], segs, 0.01)
        acc = np.zeros((H, W, 4), np.float32)
        t1 = time.perf_counter()
        st = O.pt_pass(otree, c, W, H, P, O.randseeds(1)[0], 1, acc, nthreads=cores)
        dt = time.perf_counter() - t1
        cpu_baseline = {"value": round(st.rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                        "sample": "1 full pass of the same workload (%dx%d, depth %d, seed pass 0): " % (W, H, segs) +
                                  "%d rays in %.2f s; strict-fp32 CPU restatement, %d threads" % (st.rays, dt, cores),
                        "ms_per_frame": round(dt * 1e3, 1)}
    glsl = LLVMPIPE_CONTAINER.get(args.workload) if (W, H) == (1920, 1080) else None

    out = {
        "metric": "Mrays/s (BVH queries EXECUTED by the timed fast mode: camera + bounce + Sun-shadow rays, device-counted; the "
                  "reference-defined count is in mrays_reference_defined_per_s), path tracing, 1 path/pixel/pass",
        "value": round(exe[0] / elapsed / 1e6, 3),
        "value_definition": BL.VALUE_DEFINITION,
        "metric_version": BL.METRIC_VERSION,
        "metric_version_note": "1 (rounds 1-2): rays as the reference defines them; 2 (round 3): rays the fast mode executes; 3 (round 4): the same "
                               "count, walked nearest-child-first (fewer box tests per ray); 4 (round 5): the same count, every walk in the reference's order "
                               "(the nearest-first walk became opt-in: `nearest_first_opt_in`). `value` is not comparable across versions — "
                               "`ms_per_step` is, and `mrays_reference_defined_per_s` keeps version 1's definition",
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": round(ms_step, 4),
        "ms_per_step_spread": spread,
        "repeats": len(reps),
        "repeats_note": "the K-pass timed sequence was run `repeats` times, each between barrier + synchronize; ms_per_step (and every rate) "
                        "is the median repetition, ms_per_step_spread its fastest and slowest one",
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (args.workload if (W, H) == (1920, 1080) else args.workload + " at another frame size") + ": " +
                               WORKLOADS[args.workload]["text"] + ", %dx%d, path tracing depth %d (MAX_PATH_SEGMENTS=%d, MIN_WEIGHT=0.01), "
                               % (W, H, segs, segs) + "1 path/pixel/pass, Sun direct lighting on, " + WORKLOADS[args.workload]["camera"],
                   "frame": [W, H], "parallelism": "8-row screen bands interleaved over %d rank(s)" % world,
                   "gather": None if world == 1 else (gather_note or "gpuart_hip_gather (RCCL send/recv to rank 0 + row scatter), in the timed region"),
                   "bvh_nodes": info["nodes"], "bvh_primitives": info["prims"], "bvh_depth": info["max_depth"],
                   "scene_device_bytes": info["device_bytes"], "scene_setup_s": round(setup_s, 3)},
        "warmup_note": "before the W warm-up passes: the untimed work-counting replays and one untimed K-pass sequence of the timed shape "
                       "(first use of every pass lane costs ~3 ms once)",
        "ms_per_frame": round(ms_step, 4),
        "ms_per_frame_note": "throughput figure: wall time of the K passes / K, with up to 64 passes in flight between two observations",
        "ms_per_frame_single": None if single_ms is None else round(single_ms, 4),
        "ms_per_frame_single_note": "one pass submitted and observed alone (frame time of the reference's interactive loop), median of 5",
        "nearest_first_opt_in": nearest,
        "mpaths_per_s": round(W * H * K / elapsed / 1e6, 3),
        "rays_per_step": rays / K,
        "rays_executed_per_step": exe[0] / K,
        "mrays_executed_per_s": round(exe[0] / elapsed / 1e6, 3),
        "mrays_reference_defined_per_s": round(rays / elapsed / 1e6, 3),
        "mrays_reference_defined_note": "every closest-hit query the REFERENCE performs for these passes (mode-1 replay) over the same wall "
                                        "time: what the images are worth in the reference's own work, not work this run performed",
        "frames_verified": frames_verified,
        "frames_verified_note": ("after the last timed repetition every rank's accumulator (its K timed passes) == the accumulator of the untimed mode-1 "
                                 "replay (reference-order walks, every reference query) and of the mode-4 replay (the one that counts `value`'s rays) of "
                                 "the same passes and seeds, bit for bit" if frames_verified else
                                 "MISMATCH: %d pixels differ from the mode-1 replay, %d from the mode-4 replay (summed over ranks): this line is not a measurement" % tuple(frames_bad)),
        "frames_xor_of_rank0_accumulator_bits": "%08x" % rows_crc,
        "multi_gpu": multi,
        "nodes_per_step": nodes / K, "nodes_executed_per_step": exe[1] / K,
        "rewalked_queries_per_step": exe[8] / K,
        "rewalked_queries_note": "closest-hit queries whose nearest-child-first walk could not certify its answer (a loose winner with a runner-up "
                                 "within the band, an odd box: device_scene.h) and that were walked again in the reference's order; their second walk "
                                 "is part of nodes_executed_per_step",
        "segments_per_step": segments / K,
        "algorithmic_bytes_per_ray": round(alg_bytes / rays, 1) if rays else None,
        "roofline": roof,
        "cpu_baseline": cpu_baseline,
        "cpu_baseline_reference_glsl": None if glsl is None else {
            "value": glsl["value"], "unit": "Mrays/s", "cores": 8, "kind": "reference", "ms_per_frame": glsl["ms_per_frame"],
            "where": "BUILD CONTAINER (8 vCPU Xeon), not this box: the reference's unmodified GLSL on Mesa llvmpipe, same scene / camera / "
                     "ray definition (tests/golden/time_llvmpipe.py; profiles/r02/llvmpipe_reference_glsl_container.txt, profiles/r03/llvmpipe_reference_glsl_container_871k.txt)"},
    }
    print(json.dumps(out), flush=True)
    r.close()
    shutil.rmtree(tmpdir, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()
    if not frames_verified:
        sys.exit("bench.py: the timed frames differ from the replays' (frames_verified_note)")
    if gather_verified is False:
        sys.exit("bench.py: the gathered frame differs from rank 0's own whole-frame render (multi_gpu.gather_verified_note)")


class _P2P:
    """torch.distributed send/recv for host tensors over whatever backend is up (gloo: as is; nccl: staged through the device)."""

    def __init__(self, dist, xdev):
        self.dist, self.xdev = dist, xdev

    def send(self, t, dst):
        self.dist.send(t.to(self.xdev), dst=dst)

    def recv(self, t, src):
        buf = t.to(self.xdev)
        self.dist.recv(buf, src=src)
        t.copy_(buf)


if __name__ == "__main__":
    main()
