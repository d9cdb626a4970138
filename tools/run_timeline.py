#!/usr/bin/env python3
"""Per-wave timeline of one k_run launch (a library built with -DGD_RUN_TIMELINE, tools/ab_build.sh): when the waves start,
when the path cursor runs dry, when they end, and how full their lanes were.   GPUART_LIBDIR=... python3 tools/run_timeline.py [K]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H = 1920, 1080
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(B.make_prims(S.scene_d()))
r.set_max_path_segments(8)
r.backend.set_mode(5)
SHARE = int(os.environ.get("TL_SHARE", "1"))  # TL_SHARE=8: rank 0's share of the frame among 8 ranks (what an N = 8 rank runs)
if SHARE > 1:
    from gpuart_amd import sharding
    y0, rows, band, stride, _ = sharding.interleaved_rows(0, SHARE, H)
    assert r.set_interleaved_tile(0, y0, W, rows, band, stride)
L = B.hip_lib()
L.gpuart_hip_debug_run_hist.argtypes = [C.c_void_p, C.c_void_p]
hist = np.zeros(256, np.uint64)
for it in range(3):
    if it == 2:
        L.gpuart_hip_debug_run_hist(r.backend.ctx, hist.ctypes.data_as(C.c_void_p))  # clear
    r.restart_path_tracing(1, K)
    for _ in range(K):
        r.path_tracing_pass()
    r.finish()
ev = np.zeros(4, np.uint64)
L.gpuart_hip_debug_stack_events.argtypes = [C.c_void_p, C.c_void_p]
L.gpuart_hip_debug_stack_events(r.backend.ctx, ev.ctypes.data_as(C.c_void_p))
print("traversal stack, all launches of this process: %d pushes, %d of them spill an entry to global memory (%.1f %%); %d pops, %d reload one (%.1f %%)"
      % (ev[0], ev[1], 100.0 * ev[1] / max(1, ev[0]), ev[2], ev[3], 100.0 * ev[3] / max(1, ev[2])))
n = 4096
buf = np.zeros((n, 24), np.uint64)
L.gpuart_hip_debug_run_timeline.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
rc = L.gpuart_hip_debug_run_timeline(r.backend.ctx, buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0, rc
buf = buf[buf[:, 2] > 0]
t0 = buf[:, 0].min()
us = lambda x: (x.astype(np.float64) - float(t0)) / 100.0
st, dry, end = us(buf[:, 0]), us(buf[:, 1]), us(buf[:, 2])
q = lambda a: " ".join("%7.0f" % v for v in np.percentile(a, [0, 10, 50, 90, 100]))
print("%d waves; microseconds after the first wave's start, percentiles 0/10/50/90/100" % len(buf))
print("start      ", q(st))
print("cursor dry ", q(dry))
print("end        ", q(end))
print("tail (end - dry)", q(end - dry))
for name, col, rc in (("pairs", 12, 14), ("quads", 13, 15)):
    sel = buf[:, col] > 0
    if sel.any():
        print("to %s     " % name, q(us(buf[sel, col])), " (%d waves; rounds in that mode, median %d)" % (int(sel.sum()), int(np.median(buf[sel, rc]))))
f = lambda k, m=slice(None): buf[m, k].astype(np.float64).sum()
if f(15) > 0:
    for name, m in (("all waves", slice(None)), ("the last 5 % to end", end >= np.percentile(end, 95))):
        r4 = f(15, m)
        print("quad rounds of %s (shader-clock cycles, three clock reads of ~100 each per round included): %d rounds of %.0f cycles with %.1f rays; "
              "box steps in %.0f %% of them, %.0f cycles each; leaf steps in %.0f %%, %.0f cycles each; the rest of a round (ballots, exit tests) %.0f"
              % (name, r4, f(20, m) / r4, f(21, m) / r4, 100 * f(17, m) / r4, f(16, m) / max(1, f(17, m)), 100 * f(19, m) / r4, f(18, m) / max(1, f(19, m)),
                 (f(20, m) - f(16, m) - f(18, m)) / r4))
sel = (buf[:, 13] > 0) & (end >= np.percentile(end, 95))
if sel.any():
    dq = (end[sel] - us(buf[sel, 13])); rq = buf[sel, 15].astype(np.float64)
    print("the last 5 %% of the waves to end (%d): %.0f us in quad mode (median), %d rounds there, %.2f us per round (retire / shade episodes included)"
          % (int(sel.sum()), np.median(dq), int(np.median(rq)), float(dq.sum() / max(1.0, rq.sum()))))
sel = (buf[:, 13] > 0) & (end <= np.percentile(end, 30))
if sel.any():
    dq = (end[sel] - us(buf[sel, 13])); rq = buf[sel, 15].astype(np.float64)
    print("the first 30 %% of the waves to end (%d): %.0f us in quad mode (median), %d rounds there, %.2f us per round"
          % (int(sel.sum()), np.median(dq), int(np.median(rq)), float(dq.sum() / max(1.0, rq.sum()))))
print("ready / shade list lengths summed over the rounds before the cursor ran dry / all rounds (lower bound of the mean): %.1f / %.1f" % (buf[:, 6].sum() / buf[:, 4].sum(), buf[:, 7].sum() / buf[:, 4].sum()))
tt, to, tr, tg = [buf[:, k].astype(np.float64) for k in (8, 9, 10, 11)]
print("after the cursor ran dry, per wave (medians): %.0f us in traversal rounds (%d rounds, %.2f us each), %.0f us in retire / shade / refill (%d times, %.2f us each)" % (
    np.median(tt) / 100, int(np.median(tr)), tt.sum() / max(1, tr.sum()) / 100, np.median(to) / 100, int(np.median(tg)), to.sum() / max(1, tg.sum()) / 100))
print("mean busy lanes per traversal round: %.1f of 64; rounds per wave (median) %d" % (buf[:, 3].sum() / buf[:, 4].sum(), int(np.median(buf[:, 4]))))
# lane occupancy over time is not recorded; the histogram of wave ends says how the machine empties
h, edges = np.histogram(end, bins=12)
print("wave ends per interval:", " ".join("%d@%.0f" % (c, e) for c, e in zip(h, edges[1:])))
L.gpuart_hip_debug_run_hist(r.backend.ctx, hist.ctypes.data_as(C.c_void_p))
print("per 25 us since each wave's own start (all start within 3 us): busy lanes (of %d), running waves" % (64 * len(buf)))
for k in range(0, 128, 4):
    lt = hist[k:k + 4].astype(np.float64).sum() / (4 * 2500.0); wt = hist[128 + k:128 + k + 4].astype(np.float64).sum() / (4 * 2500.0)
    if wt > 0:
        print("  %4d us  %7.0f lanes  %5.0f waves  %4.1f lanes per running wave" % (k * 25, lt, wt, lt / wt))
hw = buf[:, 5]
xcc = (hw >> np.uint64(32)) & np.uint64(0xf)
hid = hw & np.uint64(0xffffffff)
cu = (xcc << np.uint64(8)) | (((hid >> np.uint64(13)) & np.uint64(7)) << np.uint64(5)) | (((hid >> np.uint64(12)) & np.uint64(1)) << np.uint64(4)) | ((hid >> np.uint64(8)) & np.uint64(15))
ids = np.unique(cu)
last = np.array([end[cu == i].max() for i in ids]); first = np.array([end[cu == i].min() for i in ids]); med = np.array([np.median(end[cu == i]) for i in ids])
print("%d CUs seen (waves per CU: %s)" % (len(ids), q(np.array([(cu == i).sum() for i in ids]))))
print("per CU: first wave end", q(first)); print("per CU: median wave end", q(med)); print("per CU: last wave end  ", q(last))
for x in np.unique(xcc):
    print("XCC %d: waves %d, last end %.0f, mean end %.0f" % (int(x), int((xcc == x).sum()), end[xcc == x].max(), end[xcc == x].mean()))
r.close()
