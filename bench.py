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

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L1_PEAK_GACC = 1010.0           # the highest vector-L1 (TCP) cache-access rate tools/ubench reaches on the box with the product's own node fetch
                                # (2 x dwordx4 + 2 x dwordx3 per lane, every lane its own record, records resident in the L1): 1.00-1.01e12/s, flat
                                # from 4 to 8 waves per SIMD; 8.65-8.75e11 when 5-20 % of the accesses miss to L2 (profiles/r02/l1_access_calibration.txt).
                                # One access per cycle and CU would be 614.4.
STEP_PEAK_GVISITS = 227.0       # 64 lanes x 3.55e9 wave-steps/s: the rate at which tools/ubench (k_step) performs the product's traversal step — fetch one
                                # 64-byte record per lane (every lane its own, L1-resident) + the two aabb_entry tests on it — at k_trace's 6 waves per SIMD
                                # (8 waves: 3.73e9; fetch alone 4.05e9, tests alone 4.87e9: the hardware overlaps them to 1.14 x the slower one)
VALU_PEAK_GINSTR = 1228.8       # 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (same guide, "Wave scheduling")
VALU_MEASURED_GINSTR = 1058.0   # the highest issue rate tools/ubench reaches on the box: independent 4-byte v_add_f32, 128 between two branches, 8 waves per
                                # SIMD = 2.32 cycles (6 waves: 2.44; 8-byte v_fma_f32: 2.54-2.72; profiles/r02/ubench.txt)
VALU_SAME_MIX_GINSTR = 846.0    # the product's own box test on registers (87 VALU + 17 SALU per test: selects, dependent chains) at k_trace's 6 waves per SIMD:
                                # 2.90 cycles per VALU instruction

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
    ap.add_argument("--gather-timeout", type=float, default=120.0,
                    help="N > 1: seconds a rank waits for the frame gather before it names the ranks that have not arrived and exits non-zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the rocprofv3 child runs (roofline counters become null)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="rank 0 also renders the whole frame alone and checks the gathered frame against it bit for bit")
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


def profile_children(args, K, Wm):
    """rocprofv3 child runs of `bench.py --render-only` (the same W + K passes): SQ_INSTS_VALU etc. in one --pmc pass, FETCH_SIZE and
    WRITE_SIZE in one pass each (the TCC slots do not hold both, MI355X_MICROARCH.md "rocprofv3 PMC slots"). Counters are summed over
    every dispatch of the child and divided by its W + K passes. Returns (dict, note)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.abspath(__file__), "--render-only", "--steps", str(K), "--warmup", str(Wm), "--workload", args.workload,
             "--frame", args.frame]
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for tag, counters in (("valu", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES"]),
                          ("tcp", ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TOTAL_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"]),
                          ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum"]),
                          ("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"])):
        d = tempfile.mkdtemp(prefix="gpuart_pmc_", dir="/tmp")
        try:
            p = subprocess.run([exe, "--kernel-trace", "--output-format", "csv", "--pmc"] + counters + ["-d", d, "-o", "x", "--"] + child,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s failed (rc %d): %s" % (" ".join(counters), p.returncode, (p.stderr or p.stdout)[-300:])
            tot, per_kernel = {}, {}
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(path)):
                    v = float(row["Counter_Value"])
                    tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + v
                    if row["Counter_Name"] == "SQ_INSTS_VALU":
                        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                        per_kernel[k] = per_kernel.get(k, 0.0) + v
            out[tag] = (tot, per_kernel)
        except Exception as e:  # noqa: BLE001
            return None, "rocprofv3 child run failed: %r" % (e,)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, "rocprofv3 --pmc child runs of this invocation (`%s`), sums over all dispatches / %d passes" % (" ".join(child[1:]), K + Wm)


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
        r.set_seed(5489)
        run_passes(r, Wm)
        r.finish()
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
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
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
                be.comm_init(world, rank, bytes(idt.cpu().numpy().tobytes()))
            except B.HipError as e:
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
    def count(mode):
        r.set_seed(5489)
        be.set_mode(mode)
        be.counters(reset=True)
        run_passes(r, K)
        be.finish()
        c = be.counters(reset=True)
        be.set_mode(0)
        return [c.rays, c.nodes, c.prim_tests[0], c.prim_tests[1], c.prim_tests[2], c.prim_tests[3], c.segments,
                c.algorithmic_bytes() + 32 * W * th * K]

    counts = torch.tensor(count(1) + count(4), dtype=torch.float64, device=xdev)
    if dist is not None:
        dist.all_reduce(counts)
    ref, exe = counts[:8].tolist(), counts[8:].tolist()
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
            store.set("gpuart_arrived_%s_%d" % (tag, rank), "1")
        if gather_note is None:
            try:
                be.gather(1, divide_by, 0, full.data_ptr() if rank == 0 else 0)
                be.wait(int(args.gather_timeout * 1000))
            except B.HipError as e:
                if e.code == B.ERR_TIMEOUT:
                    print("bench.py rank %d: the frame gather did not complete within %.0f s (%s); ranks that never reached gather #%s: %s"
                          % (rank, args.gather_timeout, e, tag, missing_ranks(tag)), file=sys.stderr, flush=True)
                    os._exit(3)
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
    pass_ms, passes = be.kernel_time(0, reset=True)
    kernel_ms, launches = be.kernel_time(1, reset=True)
    be.set_timing(1)
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

    if dist is not None and gather_note is None and not args.verify_gather:
        be.comm_destroy()  # every rank, while all of them are still alive
    if rank != 0:
        if dist is not None:
            if args.verify_gather:
                dist.barrier()  # rank 0 is still checking the gathered frame
            dist.destroy_process_group()
        return

    if args.verify_gather and dist is not None:
        assert r.set_tile(0, 0, W, H)
        r.set_seed(5489)
        run_passes(r, K)
        whole = r.read_radiance(True)
        got = full.cpu().numpy()
        same = (got[..., :3].view(np.uint32) == whole[..., :3].view(np.uint32)).all()
        print("verify-gather: gathered %d interleaved row sets == single-rank frame: %s" % (world, bool(same)), file=sys.stderr)
        assert same, "gathered frame differs from the single-rank frame"
        dist.barrier()

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

    # ---- roofline: device-level fractions (VALU issue, vector-L1 accesses, node visits, HBM), counters from rocprofv3 child runs of this invocation ----
    ms_step = elapsed / K * 1e3
    avg_kernel_ms = kernel_ms / max(1, launches)
    roof = {"bound": "l1_accesses", "achieved": None, "peak": L1_PEAK_GACC, "unit": "G vector-L1 (TCP) cache accesses/s", "frac": None,
            "traffic": None,
            "definition": "TCP_TOTAL_CACHE_ACCESSES_sum of every kernel of a pass / ms_per_step (device level: overlapping launches are not "
                          "double counted), against the highest L1 access rate measured on the box: 1.01e12/s with the product's own node-fetch "
                          "shape on L1-resident records, flat from 4 to 8 waves per SIMD; 8.7e11/s when 5-20 % of the accesses miss to L2 "
                          "(tools/ubench; one access per cycle and CU would be 6.14e11 -> frac_of_one_access_per_clock). The BVH queries are bound by vector-memory requests: every extra 16-byte fetch per node "
                          "visit costs +12 % whether it hits L1 or not - its full service time at that peak rate - while extra VALU work costs a "
                          "third of its issue time (profiles/r02/vector_memory_bound.txt); the real mix (misses, stores, narrow requests) costs "
                          "more per access than the calibrating pattern",
            "valu_issue": {"peak": VALU_PEAK_GINSTR, "unit": "G wave64 VALU instructions/s",
                           "definition": "SQ_INSTS_VALU of every kernel of a pass / ms_per_step against 1024 SIMDs x 2.4 GHz / 2 cycles per "
                                         "wave64 VALU instruction (`peak`); `peak_measured` = the highest rate any tools/ubench loop reaches on the box "
                                         "(2.32 cycles), `peak_same_instruction_mix` = the product's own box test running on registers at 6 waves per SIMD "
                                         "(2.90 cycles per VALU instruction)"},
            "node_visits": {"peak": STEP_PEAK_GVISITS, "unit": "G node visits/s (lane level)",
                            "achieved": round(exe[1] / elapsed / 1e9, 2), "frac": round(exe[1] / elapsed / 1e9 / STEP_PEAK_GVISITS, 4),
                            "definition": "node visits the fast mode executes (device counters, mode-4 replay of the same passes) / wall time, against "
                                          "64 x the rate at which a register-resident micro-benchmark performs the same step on L1-resident records with "
                                          "every lane on its own record (tools/ubench k_step, profiles/r02/ubench_step.txt). A second opinion beside `frac`: "
                                          "the pass spends its time on traversal steps at close to the rate the chip can perform them; it is not below 1 by "
                                          "much because lanes that share a record (the top of the tree, coherent camera rays) are cheaper than the "
                                          "calibrating pattern, and it ignores leaf tests, shading and path state"},
            "kernel": "k_trace (closest-hit + Sun-shadow BVH-query launches of the wavefront pipeline)",
            "kernel_avg_ms": round(avg_kernel_ms, 5), "kernel_launches": launches, "kernel_ms_summed_per_pass": round(kernel_ms / (K * len(reps)), 4),
            "kernel_concurrency": round(kernel_ms / (elapsed_all * 1e3), 3),
            "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "algorithmic_bytes_per_pass_reference": alg_bytes / K, "algorithmic_bytes_per_pass_executed": exe[7] / K,
                    "algorithmic_rate_executed": round(exe[7] / elapsed / 1e9, 1),
                    "algorithmic_rate_over_peak": round(exe[7] / elapsed / 1e9 / HBM_PEAK_GBS, 3),
                    "note": "algorithmic bytes (SURVEY.md 8(d): 48 B per node tested + 32/48/64/80 B per primitive tested + 32 B per pixel "
                            "and pass) / wall time. Above the HBM peak because the tree is cache-resident (L2 / Infinity Cache): not a "
                            "fraction of the HBM roofline; `traffic_frac` is"}}
    if world == 1 and not args.no_profile:
        prof, note = profile_children(args, K, Wm)
        roof["source"] = note
        if prof:
            n = K + Wm
            v = prof["valu"][0]
            valu = v.get("SQ_INSTS_VALU", 0.0) / n
            vi = roof["valu_issue"]
            vi["instr_per_pass"] = valu
            vi["achieved"] = round(valu / (ms_step * 1e-3) / 1e9, 2)
            vi["frac"] = round(vi["achieved"] / VALU_PEAK_GINSTR, 4)
            vi["peak_measured"] = VALU_MEASURED_GINSTR
            vi["frac_of_measured_peak"] = round(vi["achieved"] / VALU_MEASURED_GINSTR, 4)
            vi["peak_same_instruction_mix"] = VALU_SAME_MIX_GINSTR
            vi["frac_of_same_mix_peak"] = round(vi["achieved"] / VALU_SAME_MIX_GINSTR, 4)
            if v.get("SQ_ACTIVE_INST_VALU"):
                vi["lane_util"] = round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4)
            pk = prof["valu"][1]
            tot = sum(pk.values()) or 1.0
            vi["kernel_share"] = round(sum(x for k, x in pk.items() if k.startswith("k_trace")) / tot, 4)
            vi["salu_instr_per_pass"] = v.get("SQ_INSTS_SALU", 0.0) / n
            vi["vmem_read_instr_per_pass"] = v.get("SQ_INSTS_VMEM_RD", 0.0) / n
            tc = prof["tcp"][0]
            acc = tc.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / n
            roof["l1_cache_accesses_per_pass"] = acc
            roof["l1_requests_before_coalescing_per_pass"] = tc.get("TCP_TOTAL_ACCESSES_sum", 0.0) / n
            roof["l1_misses_to_l2_per_pass"] = tc.get("TCP_TCC_READ_REQ_sum", 0.0) / n
            roof["achieved"] = round(acc / (ms_step * 1e-3) / 1e9, 2)
            roof["frac"] = round(roof["achieved"] / L1_PEAK_GACC, 4)
            roof["frac_of_one_access_per_clock"] = round(roof["achieved"] / 614.4, 4)
            l2 = prof["tcc"][0]
            if l2.get("TCC_REQ_sum"):
                roof["l2"] = {"requests_per_pass": l2["TCC_REQ_sum"] / n, "hits_per_pass": l2.get("TCC_HIT_sum", 0.0) / n,
                              "misses_per_pass": l2.get("TCC_MISS_sum", 0.0) / n,
                              "hit_rate": round(l2.get("TCC_HIT_sum", 0.0) / max(1.0, l2.get("TCC_HIT_sum", 0.0) + l2.get("TCC_MISS_sum", 0.0)), 4),
                              "note": "TCC_HIT / TCC_MISS / TCC_REQ summed over the L2 channels and all kernels of a pass; misses go on to the "
                                      "Infinity Cache and HBM (`traffic`)"}
            # HBM traffic: FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B; calibrated for wide streams — our 16-B gathers
            # are uncalibrated, read it as an upper estimate), both counters in KB; counts Infinity-Cache hits too
            fetch = prof["fetch"][0].get("FETCH_SIZE", 0.0) / n
            write = prof["write"][0].get("WRITE_SIZE", 0.0) / n
            traffic = (2.0 * fetch + write) * 1024.0
            roof["traffic"] = traffic
            roof["hbm"].update({"traffic_bytes_per_pass": traffic, "FETCH_SIZE_KB_per_pass": fetch, "WRITE_SIZE_KB_per_pass": write,
                                "traffic_rate": round(traffic / (ms_step * 1e-3) / 1e9, 1),
                                "traffic_frac": round(traffic / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
    else:
        roof["source"] = "not collected (%s)" % ("--no-profile" if args.no_profile else "N > 1: one GPU's counters would not describe the job")
    # The line's headline fraction is the VALU-issue one, priced against the guide's peak (the only one of the three views whose peak is
    # not of our own measuring); the vector-L1 view moves into `l1_accesses`, the step view stays in `node_visits`. The traversal step is
    # balanced on the L1 address stage and VALU issue, so no single fraction tells the whole story.
    l1_keys = ("achieved", "peak", "unit", "frac", "definition", "l1_cache_accesses_per_pass", "l1_requests_before_coalescing_per_pass",
               "l1_misses_to_l2_per_pass", "frac_of_one_access_per_clock")
    roof["l1_accesses"] = {k: roof.pop(k) for k in l1_keys if k in roof}
    vi = roof["valu_issue"]
    roof.update({"bound": "valu_issue", "achieved": vi.get("achieved"), "peak": vi["peak"], "unit": vi["unit"], "frac": vi.get("frac"),
                 "definition": vi["definition"] + ". Beside it: `l1_accesses` (vector-L1 cache accesses against the highest rate measured on the box) and "
                               "`node_visits` (executed node visits against the micro-benchmarked rate of the same traversal step); `traffic` = HBM "
                               "bytes per pass from the PMC counters. NOTE: `frac` is a share of ISSUE SLOTS, not of useful work — removing "
                               "wasted instructions lowers it: round 3's thin-wave modes issue 5.7e8 instead of 6.4e8 VALU instructions per "
                               "pass (same rays, same node visits) in 6 % less time, which moves `frac` from 0.56 to 0.53 while "
                               "`node_visits.frac` rises from 0.87 to 0.89-0.90"})

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
        "mpaths_per_s": round(W * H * K / elapsed / 1e6, 3),
        "rays_per_step": rays / K,
        "rays_executed_per_step": exe[0] / K,
        "mrays_executed_per_s": round(exe[0] / elapsed / 1e6, 3),
        "mrays_reference_defined_per_s": round(rays / elapsed / 1e6, 3),
        "mrays_reference_defined_note": "every closest-hit query the REFERENCE performs for these passes (mode-1 replay) over the same wall "
                                        "time: what the images are worth in the reference's own work, not work this run performed",
        "multi_gpu": multi,
        "nodes_per_step": nodes / K, "nodes_executed_per_step": exe[1] / K,
        "segments_per_step": segments / K,
        "algorithmic_bytes_per_ray": round(alg_bytes / rays, 1) if rays else None,
        "roofline": roof,
        "cpu_baseline": cpu_baseline,
        "cpu_baseline_reference_glsl": None if glsl is None else {
            "value": glsl["value"], "unit": "Mrays/s", "cores": 8, "kind": "reference", "ms_per_frame": glsl["ms_per_frame"],
            "where": "BUILD CONTAINER (8 vCPU Xeon), not this box: the reference's unmodified GLSL on Mesa llvmpipe, same scene / camera / "
                     "ray definition (tests/golden/time_llvmpipe.py; profiles/r02/llvmpipe_reference_glsl_container.txt, profiles/r03/llvmpipe_reference_glsl_container_871k.txt)"},
    }
    print(json.dumps(out))
    r.close()
    shutil.rmtree(tmpdir, ignore_errors=True)
    if dist is not None:
        dist.destroy_process_group()


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
