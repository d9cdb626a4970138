#!/usr/bin/env python3
"""How many interior-node visits (64-byte record fetches) of the bench workload fall into the top levels of the tree: the
share an LDS-resident copy of the top D levels would take off the vector-memory pipeline.   python3 tools/top_levels.py [workload]"""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys
sys.path.insert(0, %r)
from gpuart_amd import binding as B, synth_scenes as S
w = sys.argv[1]
cam = dict({"cfg3": S.BENCH_CAMERA, "cfg2": S.DEFAULT_CAMERA, "cluster": S.CLUSTER_NEAR_CAMERA, "tree": S.TREE_NEAR_CAMERA}[w]); cam["dir"] = S.camera_dir(cam)
descs = {"cfg3": S.scene_d, "cfg2": S.scene_p, "cluster": S.cluster_scene, "tree": S.tree_scene}[w]()
r = B.Renderer(1920, 1080, cam); r.set_user_sphere(S.USER_SPHERE[:3], 0.0, 0.0); r.set_primitives(B.make_prims(descs))
r.set_max_path_segments({"cfg3": 8, "cfg2": 4, "cluster": 5, "tree": 5}[w])
be = r.backend; be.set_mode(4); be.counters(reset=True)
r.restart_path_tracing(1, 4); [r.path_tracing_pass() for _ in range(4)]; r.finish()
c = be.counters()
print(c.box_steps, c.box_steps_top, c.rays)
""" % ROOT
w = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
for d in (4, 6, 8, 9, 10, 11, 12, 14):
    out = subprocess.run([sys.executable, "-c", CHILD, w], env=dict(os.environ, GPUART_HIP_TOP_DEPTH=str(d)), capture_output=True, text=True, check=True).stdout.split()
    steps, top, rays = (int(x) for x in out)
    print("%s: top %2d levels (<= %5d records, %4d KB): %.1f %% of %.1f node visits per ray" % (w, d, 2 ** d - 1, (2 ** d - 1) * 64 // 1024, 100.0 * top / steps, steps / rays))
