#!/usr/bin/env python3
"""A Sun on the horizon (altitude 0: the Sun direction's z component is exactly 0) makes every Sun-shadow ray one the quick box answers
do not vet (csrc/hip/box_quick.h gq_ray_slack: a direction component of exactly 0): ms per pass of cfg3 with such a Sun, for the
environment's GPUART_HIP_QUICK_BOXES setting.   python3 tools/time_sunset.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(1920, 1080, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
r.set_primitives(B.make_prims(S.scene_d()))
r.set_max_path_segments(8)
for alt, what in ((0.0, "Sun on the horizon (altitude 0: shadow rays with d.z == 0)"), (S.SUN_ALTITUDE, "default Sun")):
    r.set_sun(S.SUN_AZIMUTH, alt, True)
    best = 1e9
    for rep in range(4):
        r.set_seed(5489); r.restart_path_tracing(1, K)
        t0 = time.perf_counter()
        for _ in range(K):
            r.path_tracing_pass()
        r.finish()
        if rep:
            best = min(best, (time.perf_counter() - t0) / K * 1e3)
    print("%-62s QUICK_BOXES=%s  %.3f ms/pass" % (what, os.environ.get("GPUART_HIP_QUICK_BOXES", "1"), best))
r.close()
