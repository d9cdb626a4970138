#!/usr/bin/env python3
"""Where the host BVH build of the 871 200-triangle mesh spends its time (GPUART_HOST_TIMING=2: the constructor's phases and, per node of
at least 65 536 items, box / keys / sort / permute / split).   python3 tools/bvh_build_phases.py [big|scene_d] [threads]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %r)
from gpuart_amd import binding as B, synth_scenes as S
which = sys.argv[1]
descs = S.scene_d() if which == "scene_d" else S.scene_d(660, 660)
p = B.make_prims(descs)
for i in range(3):
    t = time.perf_counter(); q, depth = B.compile_bvh(p); print("== build + compile %%.1f ms" %% ((time.perf_counter() - t) * 1e3), file=sys.stderr, flush=True)
""" % ROOT
env = dict(os.environ, GPUART_HOST_TIMING="2")
if len(sys.argv) > 2:
    env["GPUART_BVH_THREADS"] = sys.argv[2]
r = subprocess.run([sys.executable, "-c", CHILD, sys.argv[1] if len(sys.argv) > 1 else "big"], env=env, capture_output=True, text=True)
lines = r.stderr.strip().split("\n")
# the last of the three builds (warm)
last = max(i for i, l in enumerate(lines[:-1]) if l.startswith("== ")) if sum(l.startswith("== ") for l in lines) > 1 else -1
print("\n".join(lines[last + 1:]))
