#!/usr/bin/env python3
"""Times the reference's unmodified GLSL on Mesa llvmpipe (container-only: needs /root/reference/shaders and
oracle/_ref/libglref.so) on the bench workload, for the "reference GLSL path under llvmpipe" baseline of BASELINE.md.

    python tests/golden/time_llvmpipe.py [--width 1920 --height 1080 --passes 3 --scene scene_d|box]
Prints one JSON line: ms per path-tracing pass (1 path/pixel, MAX_PATH_SEGMENTS=8) and per direct-lighting frame, with the
exact ray count of the same pass from the oracle's counters -> Mrays/s."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "glref"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import glref  # noqa: E402
import make_golden as MG  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--scene", default="scene_d")
ap.add_argument("--max-segments", type=int, default=8)
a = ap.parse_args()
if not glref.available():
    sys.exit("needs /root/reference/shaders and oracle/_ref/libglref.so (make -C oracle)")
gl = glref.GLRef()
W, H = a.width, a.height
_, tree, _ = MG.scene_tree(a.scene)
cam = MG.default_cam(S.BENCH_CAMERA if a.scene in ("scene_d", "dragon871k") else S.DEFAULT_CAMERA)
r = MG.RefRenderer(gl, MG.RefPrograms(gl, a.max_segments), W, H, cam, tree)
seeds = O.randseeds(a.passes + 1)
r.direct()
t0 = time.perf_counter(); r.direct(); t_direct = time.perf_counter() - t0
r.reset(); r.pt_pass(1, seeds[0])  # warm-up (shader JIT)
times = []
for k in range(a.passes):
    t0 = time.perf_counter(); r.pt_pass(1, seeds[k + 1]); times.append(time.perf_counter() - t0)
# exact ray count of one such pass (oracle counters; identical work definition to bench.py)
sun = O.sun_direction(S.SUN_AZIMUTH, S.SUN_ALTITUDE)
P = O.make_params(sun, S.SUN_ALTITUDE, True, S.USER_SPHERE, 0.0, 0, float(r.cam[12]), r.cam[0:3], a.max_segments, 0.01)
acc = np.zeros((H, W, 4), np.float32)
st = O.pt_pass(tree, r.cam, W, H, P, seeds[1], 1, acc, nthreads=os.cpu_count())
med = float(np.median(times))
print(json.dumps({"renderer": gl.renderer(), "lp_num_threads": os.environ.get("LP_NUM_THREADS", "default"), "cores": os.cpu_count(),
                  "scene": a.scene, "frame": [W, H], "max_segments": a.max_segments, "pt_ms_per_pass": round(med * 1e3, 1),
                  "direct_ms_per_frame": round(t_direct * 1e3, 1), "rays_per_pass": st.rays,
                  "mrays_per_s": round(st.rays / med / 1e6, 3), "mpaths_per_s": round(W * H / med / 1e6, 3)}))
