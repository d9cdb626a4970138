#!/usr/bin/env python3
"""Quick box answers (csrc/hip/box_quick.h) against the six face tests ON THE DEVICE: a library built with -DGD_QUICK_CHECK
(tools/ab_build.sh qcheck "-DGD_QUICK_CHECK") runs every fast-form box test of the wide traversal steps both ways and counts.
Renders K passes of each workload in the timed mode (0), the k_run mode (5) and direct lighting.
    GPUART_LIBDIR=gpuart_amd/lib_ab/qcheck python3 tools/quick_box_stats.py [K] [workload ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WORK = sys.argv[2:] or ["cfg3", "cfg2", "dragon871k", "cluster", "tree", "box"]
L = B.hip_lib()
L.gpuart_hip_debug_quick_stats.argtypes = [C.c_void_p, C.c_void_p]
total_bad = 0


def stats(r):
    ev = np.zeros(8, np.uint64)
    rc = L.gpuart_hip_debug_quick_stats(r.backend.ctx, ev.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return [int(v) for v in ev]


for w in WORK:
    cam = dict({"cfg2": S.DEFAULT_CAMERA, "box": S.DEFAULT_CAMERA, "cluster": S.CLUSTER_NEAR_CAMERA, "tree": S.TREE_NEAR_CAMERA}.get(w, S.BENCH_CAMERA)); cam["dir"] = S.camera_dir(cam)
    r = B.Renderer(1920, 1080, cam)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    if w == "box":
        r.init_box()
    elif w == "cluster":
        r.set_primitives(B.make_prims(S.cluster_scene()))
    elif w == "tree":
        r.set_primitives(B.make_prims(S.tree_scene()))
    else:
        r.set_primitives(B.make_prims(S.scene_d(660, 660) if w == "dragon871k" else S.scene_p() if w == "cfg2" else S.scene_d()))
    r.set_max_path_segments(4 if w == "cfg2" else 5 if w in ("cluster", "tree") else 8)
    stats(r)
    for mode, name in ((0, "pipeline"), (5, "k_run"), (-1, "direct")):
        if mode >= 0:
            r.backend.set_mode(mode)
            r.restart_path_tracing(1, K)
            for _ in range(K):
                r.path_tracing_pass()
            r.finish()
        else:
            r.backend.set_mode(0)
            r.render_direct()
            r.read_direct()
        ev = stats(r)
        total_bad += ev[4]
        print("%-11s %-9s boxes %.4g  stand %.6f %%  steps %.4g  steps with a withdrawn lane %.4f %%  MISMATCHES %d (odd among them %d)" % (
            w, name, ev[0], 100.0 * ev[1] / max(1, ev[0]), ev[2], 100.0 * ev[3] / max(1, ev[2]), ev[4], ev[5]), flush=True)
    r.close()
print("total mismatches:", total_bad)
sys.exit(1 if total_bad else 0)
