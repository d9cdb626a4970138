#!/usr/bin/env python3
"""Per-pass time of rank 0's share of the frame (TILE_FRAME, default 1920x1080) for N = 1, 2, 4, 8 interleaved ranks, on one GPU:
shows the fixed per-pass cost that limits strong scaling (ideal: t(N) = t(1)/N). TILE_K = passes per sequence (default 64),
TILE_N = the rank counts. BASELINE.json's cfg5 (7680x4320, 256 spp, 8 ranks): TILE_FRAME=7680x4320 TILE_K=256 TILE_N=1,8."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B, sharding, synth_scenes as S
W, H = [int(x) for x in os.environ.get("TILE_FRAME", "1920x1080").split("x")]
K = int(os.environ.get("TILE_K", "64"))
REPS = int(os.environ.get("TILE_REPS", "4"))
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(B.make_prims(S.scene_d()))
r.set_max_path_segments(8)
r.backend.set_timing(0)
t1 = None
for n in [int(x) for x in os.environ.get("TILE_N", "1,2,4,8").split(",")]:
    worst = 0
    for rank in sorted(set([0, n - 1])):
        if n > 1:
            y0, rows, band, stride, _ = sharding.interleaved_rows(rank, n, H)
            assert r.set_interleaved_tile(0, y0, W, rows, band, stride)
        # one untimed sequence of the timed shape first (the first use of every pass lane costs ~3 ms once: bench.py warms up the same
        # way), then the best of three timed ones
        best = 1e9
        for rep in range(REPS):
            r.set_seed(5489); r.restart_path_tracing(1, K)
            t0 = time.perf_counter()
            for _ in range(K): r.path_tracing_pass()
            r.finish()
            if rep: best = min(best, (time.perf_counter() - t0) / K * 1e3)
        worst = max(worst, best)
    t1 = t1 or worst
    print("%dx%d, %d passes: " % (W, H, K), end="")
    print("N=%d  %.3f ms/pass (slowest of first/last rank)  speed-up %.2fx  efficiency %.0f%%" % (n, worst, t1 / worst, 100 * t1 / worst / n), flush=True)
