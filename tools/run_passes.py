#!/usr/bin/env python3
"""Renders K path-tracing passes of the bench workload (cfg3) with the GPUART_HIP_* knobs of the environment —
the program to put behind `rocprofv3 ... --` when profiling one configuration.   python3 tools/run_passes.py [K]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
W, H = [int(x) for x in os.environ.get("FRAME", "1920x1080").split("x")]  # FRAME=3840x2160
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
r = B.Renderer(W, H, cam)
r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0)
WORKLOAD = os.environ.get("WORKLOAD", "cfg3")  # cfg3 | dragon871k (the tree that leaves the L2s) | cfg2 | cluster | tree
if WORKLOAD == "cfg2":
    cam = dict(S.DEFAULT_CAMERA); cam["dir"] = S.camera_dir(cam)
    r.set_camera(cam)
if WORKLOAD in ("cluster", "tree"):  # the reference's primitive-list scenes on their stand-ins, near cameras, depth 5 (as bench.py --workload)
    cam = dict(S.NEAR_CAMERAS[WORKLOAD]); cam["dir"] = S.camera_dir(cam)
    r.set_camera(cam)
r.set_primitives(B.make_prims(S.scene_d(660, 660) if WORKLOAD == "dragon871k" else S.scene_p() if WORKLOAD == "cfg2" else
                              S.cluster_scene() if WORKLOAD == "cluster" else S.tree_scene() if WORKLOAD == "tree" else S.scene_d()))
r.set_max_path_segments(4 if WORKLOAD == "cfg2" else 5 if WORKLOAD in ("cluster", "tree") else 8)
if int(os.environ.get("SHARE", "1")) > 1:  # SHARE=N: rank 0's share of the frame among N ranks (interleaved 8-row bands, as bench.py --gpus N)
    from gpuart_amd import sharding
    y0, rows, band, stride, _ = sharding.interleaved_rows(0, int(os.environ["SHARE"]), H)
    assert r.set_interleaved_tile(0, y0, W, rows, band, stride)
r.backend.set_mode(int(os.environ.get('GPUART_MODE', '0')))
r.backend.set_timing(int(os.environ.get('GPUART_TIMING', '0')))
import numpy as np  # noqa: E402
for _ in range(REPS):
    r.set_seed(5489)  # the same RandSeeds in every repetition and every process: the frames of two builds can be compared
    r.restart_path_tracing(1, K)
    t0 = time.perf_counter()
    for _ in range(K):
        r.path_tracing_pass()
    r.finish()
    print("%d passes, %.3f ms/pass" % (K, (time.perf_counter() - t0) / K * 1e3))
acc = r.read_radiance(False)
print("frame %08x" % int(np.bitwise_xor.reduce((acc.view(np.uint32) * np.arange(1, acc.size + 1, dtype=np.uint32).reshape(acc.shape)).ravel())))
r.close()
