#!/usr/bin/env python3
"""How full k_trace's steps are, on a library built with -DGD_STEP_STATS (tools/ab_build.sh stats "-DGD_STEP_STATS"):
box steps and the lanes that take part in them, leaf steps and theirs, rounds of the wide loop and the lanes that hold a ray in them.
    GPUART_LIBDIR=gpuart_amd/lib_ab/stats python3 tools/step_stats.py [K] [workload ...]      (knobs through GPUART_HIP_* as usual)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
WORK = sys.argv[2:] or ["cfg3"]
L = B.hip_lib()
L.gpuart_hip_debug_step_stats.argtypes = [C.c_void_p, C.c_void_p]


def stats(r):
    ev = np.zeros(16, np.uint64)
    assert L.gpuart_hip_debug_step_stats(r.backend.ctx, ev.ctypes.data_as(C.c_void_p)) == 0
    return [int(v) for v in ev]


for w in WORK:
    cam = dict({"cfg2": S.DEFAULT_CAMERA}.get(w, S.BENCH_CAMERA)); cam["dir"] = S.camera_dir(cam)
    r = B.Renderer(1920, 1080, cam)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    r.set_primitives(B.make_prims(S.scene_p() if w == "cfg2" else S.scene_d(660, 660) if w == "dragon871k" else S.scene_d()))
    r.set_max_path_segments(4 if w == "cfg2" else 8)
    r.backend.set_mode(3)  # the launch pipeline (k_trace) whatever K
    stats(r)
    r.restart_path_tracing(1, K)
    for _ in range(K):
        r.path_tracing_pass()
    r.finish()
    st = stats(r)
    box, box_lanes, leaf, leaf_lanes, rounds, held, refills = st[:7]
    trips, trip_lanes, calls, call_lanes = st[8:12]
    t_refill, t_trav, t_retire = st[12:15]
    print("%s, %d passes through k_trace:" % (w, K))
    print("  box steps  %12d   lanes per box step  %5.1f of 64" % (box, box_lanes / max(1, box)))
    print("  leaf steps %12d   lanes per leaf step %5.1f of 64   (one leaf step per %.2f box steps)" % (leaf, leaf_lanes / max(1, leaf), box / max(1, leaf)))
    print("  rounds     %12d   lanes holding a ray %5.1f of 64   refill episodes %d (one per %.1f rounds)" % (rounds, held / max(1, rounds), refills, rounds / max(1, refills)))
    print("  pops (trav_pop, all kernels of the run): %d wave-level calls with %.1f lanes each; the loop runs %.2f trips per call, %.1f lanes per trip" % (
        calls, call_lanes / max(1, calls), trips / max(1, calls), trip_lanes / max(1, trips)))
    tt = max(1, t_refill + t_trav + t_retire)
    print("  wave time by phase of k_trace's outer loop: refill %.1f %%, traverse %.1f %%, settle + retire %.1f %%   (refill episode: %.0f ticks; a round of the traverse loop: %.0f ticks of the 100 MHz clock)" % (
        100.0 * t_refill / tt, 100.0 * t_trav / tt, 100.0 * t_retire / tt, t_refill / max(1, refills), t_trav / max(1, rounds)))
    r.close()
