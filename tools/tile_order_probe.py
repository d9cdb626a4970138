#!/usr/bin/env python3
"""Probe: does the order in which the paths of a pass are born matter for the time of a short sequence? The 8x8 pixel blocks of the
tile, row-major (as shipped) against sorted by an estimate of their cost, most expensive first (camera rays through the traversal
hook: a pixel that misses = 1, that hits a triangle = 5, anything else = 2). Accumulators must be equal bit for bit.
    python3 tools/tile_order_probe.py [WORKLOAD]     (TILE_N=1,8  K=1,2,20,64)"""
import os
os.environ.setdefault("GPUART_LIBDIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuart_amd", "lib_test"))  # uses test hooks (include/gpuart_hip_test.h)
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B, sharding, synth_scenes as S  # noqa: E402

W, H = 1920, 1080
WORKLOAD = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cam = dict(S.DEFAULT_CAMERA if WORKLOAD == "cfg2" else S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(B.make_prims(S.scene_d(660, 660) if WORKLOAD == "dragon871k" else S.scene_p() if WORKLOAD == "cfg2" else S.scene_d()))
r.set_max_path_segments(4 if WORKLOAD == "cfg2" else 8)
be = r.backend
be.set_timing(0)


def cost_order(kind):
    g = be.get_share()
    tw, th = g.tw, g.th
    rs, rd = be.test_cam_rays()
    a0, a1 = be.test_traverse(rs.reshape(-1, 4), rd.reshape(-1, 4), (0.0, 0.0, 0.0, 0.0))
    typ = a1[:, 3].reshape(th, tw)
    cost = np.where(typ < 0, 1, np.where(typ == 2, 5, 2)).astype(np.int64)   # P_TRIANGLE == 2
    ty, tx = (th + 7) // 8, (tw + 7) // 8
    pad = np.zeros((ty * 8, tx * 8), np.int64); pad[:th, :tw] = cost
    tc = pad.reshape(ty, 8, tx, 8).sum((1, 3)).reshape(-1)
    if kind == "identity":
        return np.arange(tc.size, dtype=np.uint32)
    if kind == "heavy_first":
        return np.argsort(-tc, kind="stable").astype(np.uint32)
    if kind == "light_first":
        return np.argsort(tc, kind="stable").astype(np.uint32)
    if kind == "random":
        return np.random.RandomState(1).permutation(tc.size).astype(np.uint32)
    raise ValueError(kind)


def timed(K, reps=5):
    best = 1e9
    for rep in range(reps):
        r.set_seed(5489); r.restart_path_tracing(1, K)
        t0 = time.perf_counter()
        for _ in range(K):
            r.path_tracing_pass()
        r.finish()
        if rep:
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
    return best, r.read_radiance()


for n in [int(x) for x in os.environ.get("TILE_N", "1,8").split(",")]:
    if n > 1:
        y0, rows, band, stride, _ = sharding.interleaved_rows(0, n, H)
        assert r.set_interleaved_tile(0, y0, W, rows, band, stride)
    g = be.get_share()
    orders = {"row-major": np.arange(((g.tw + 7) // 8) * ((g.th + 7) // 8), dtype=np.uint32)}
    for kind in os.environ.get("ORDERS", "heavy_first").split(","):
        orders[kind] = cost_order(kind)
    orders["auto"] = None   # the library's own: the first run counts shaded segments per block, the following ones are sorted by them
    for K in [int(x) for x in os.environ.get("K", "1,2,20,64").split(",")]:
        ref = None
        line = "N=%d K=%-3d" % (n, K)
        for rnd in range(2):
            for name, o in orders.items():
                be.test_tile_order(o)
                ms, img = timed(K)
                if ref is None:
                    ref = img
                assert np.array_equal(ref.view(np.uint32), img.view(np.uint32)), "%s: the accumulator differs" % name
                line += "  %s %.3f" % (name, ms)
            line += "  |"
        print(line, flush=True)
r.close()
