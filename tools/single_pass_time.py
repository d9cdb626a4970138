#!/usr/bin/env python3
"""One pass submitted and observed alone (the reference's interactive loop) at several frame sizes, launch pipeline (mode 3)
against the persistent run kernel (mode 5).   python3 tools/single_pass_time.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
prims = B.make_prims(S.scene_d())
for W, H in ((960, 540), (1920, 1080), (2560, 1440), (3840, 2160)):
    r = B.Renderer(W, H, cam)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    r.set_primitives(prims)
    r.set_max_path_segments(8)
    out = []
    for mode in (3, 5):
        r.backend.set_mode(mode)
        for K in (1, 2):
            ts = []
            for _ in range(8):
                t = time.perf_counter()
                r.restart_path_tracing(1, K)
                for _ in range(K):
                    r.path_tracing_pass()
                r.finish()
                ts.append((time.perf_counter() - t) * 1e3 / K)
            out.append("mode %d K=%d %.2f ms/pass" % (mode, K, float(np.median(ts[2:]))))
    print("%dx%d (%.1f M paths per pass): %s" % (W, H, W * H / 1e6, "; ".join(out)))
    r.close()
