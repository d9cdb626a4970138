#!/usr/bin/env python3
"""ms per direct-lighting frame (Renderer::RenderDirectLighting) of Scene D at 1080p / 4K.   python3 tools/time_direct.py"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
prims = B.make_prims(S.scene_d())
frames = [tuple(int(v) for v in a.lower().split('x')) for a in sys.argv[1:]] or [(1920, 1080), (3840, 2160)]
for W, H in frames:
    r = B.Renderer(W, H, cam)
    r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
    r.set_primitives(prims)
    for _ in range(3):
        r.render_direct()
    r.finish()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            r.render_direct()
        r.finish()
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    print("%dx%d: %.3f ms per direct-lighting frame (%.1f Mpixels/s)" % (W, H, best, W * H / best / 1e3))
    r.close()
