#!/usr/bin/env python3
"""Where Renderer::SetPrimitives spends its time: host BVH build + compile (gpuart_compile_bvh, no device),
then the whole call on the device (build + compile + re-layout + upload).   python3 tools/setprims_time.py [scene_d|big|cluster|tree]"""
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import binding as B  # noqa: E402
from gpuart_amd import synth_scenes as S  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "scene_d"
descs = {"scene_d": S.scene_d, "big": lambda: S.scene_d(660, 660), "cluster": S.cluster_scene, "tree": S.tree_scene}[which]()
prims = B.make_prims(descs)
cam = dict(S.BENCH_CAMERA); cam["dir"] = S.camera_dir(cam)
best_host = 1e9
for _ in range(4):
    t = time.perf_counter(); q, depth = B.compile_bvh(prims); best_host = min(best_host, time.perf_counter() - t)
r = B.Renderer(64, 64, cam)
r.set_primitives(prims)
best = 1e9
lib = None
for _ in range(4):
    t = time.perf_counter(); r.set_primitives(prims); r.finish(); best = min(best, time.perf_counter() - t)
    lib = min(lib or r.last_setprims_ms(), r.last_setprims_ms())
be = r.backend
t = time.perf_counter(); be.upload_bvh(q); be.finish(); up = time.perf_counter() - t
print("%s: %d primitives, %d quads (%.1f MB), depth %d, threads %s: host build + compile %.1f ms; upload (validate, re-layout, copy) %.1f ms; "
      "Renderer::SetPrimitives %.1f ms in the library (build %.1f + compile %.1f + re-layout and upload %.1f; %.1f ms with the harness that makes and deletes one "
      "object per primitive)" % (which, len(descs), len(q), q.nbytes / 1e6, depth, os.environ.get("GPUART_BVH_THREADS", "default"),
                                 best_host * 1e3, up * 1e3, lib[0], lib[1], lib[2], lib[3], best * 1e3))
r.close()
