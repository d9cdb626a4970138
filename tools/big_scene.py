#!/usr/bin/env python3
"""cfg3's workload on a denser mesh: the displaced torus at nu x nv quads (660 x 660 = 871 200 triangles, the size of the
real Stanford dragon; 76 MB of tree instead of 8.7 MB) — ms per pass, Mrays/s, build time.   python3 tools/big_scene.py [nu]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 660
W, H, K = 1920, 1080, 64
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
t0 = time.perf_counter()
prims = B.make_prims(S.scene_d(nu, nu))
t1 = time.perf_counter()
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(prims)
t2 = time.perf_counter()
r.set_max_path_segments(8)
be = r.backend
info = be.scene_info()
be.set_mode(1); be.counters(reset=True)
r.restart_path_tracing(1, 2); r.path_tracing_pass(); r.path_tracing_pass(); r.finish()
cnt = be.counters(reset=True)
be.set_mode(0); be.set_timing(0)
best = 1e9
for _ in range(3):
    r.restart_path_tracing(1, K)
    t = time.perf_counter()
    for _ in range(K):
        r.path_tracing_pass()
    r.finish()
    best = min(best, (time.perf_counter() - t) / K)
print("%d triangles: %d nodes, depth %d, %.1f MB on the device; SetPrimitives (build + compile + upload) %.2f s"
      % (2 * nu * nu, info["nodes"], info["max_depth"], info["device_bytes"] / 1e6, t2 - t1))
print("1080p depth 8: %.3f ms per pass, %.0f Mrays/s (%.2f M rays per pass, %.0f B algorithmic per ray)"
      % (best * 1e3, cnt.rays / 2 / best / 1e6, cnt.rays / 2 / 1e6, cnt.algorithmic_bytes() / cnt.rays))
r.close()
