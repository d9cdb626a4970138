#!/usr/bin/env python3
"""Times the host BVH build: one thread (GPUART_BVH_THREADS=1, the reference's algorithm as it stands:
std::sort per node) against the parallel build that is the default (halves of large nodes on different threads + the
node's own sort on several, csrc/host/exact_sort.h), in child processes so that the environment variable is read afresh;
all must give the same bytes. Reported: the time the library spends (BoundingVolumesHierarchy constructor + CompileTo,
gpuart_last_build_ms) and, beside it, the wall time of the ctypes call with its harness work.   python3 tools/bvh_build_time.py [scene_d|big|cluster]   (big = 871 200 triangles)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, hashlib
sys.path.insert(0, %r)
from gpuart_amd import binding as B, synth_scenes as S
which = sys.argv[1]
descs = S.scene_d() if which == "scene_d" else S.scene_d(660, 660) if which == "big" else S.cluster_scene()
p = B.make_prims(descs)
best = 1e9
lib = (1e9, 0.0)
for i in range(5):
    t = time.perf_counter(); q, depth = B.compile_bvh(p); best = min(best, time.perf_counter() - t)
    lib = min(lib, B.last_build_ms(), key=sum)
print("%%.1f %%s %%d %%d %%.1f %%.1f" %% (best * 1e3, hashlib.sha256(q.tobytes()).hexdigest()[:16], depth, len(descs), lib[0], lib[1]))
""" % ROOT


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "scene_d"
    out = {}
    for threads in ("1", "2", "4", "8", "16", "default"):
        env = dict(os.environ)
        if threads != "default":
            env["GPUART_BVH_THREADS"] = threads
        r = subprocess.run([sys.executable, "-c", CHILD, which], env=env, capture_output=True, text=True, check=True)
        ms, digest, depth, n, build, comp = r.stdout.split()
        out[threads] = (float(build) + float(comp), digest)
        print("%s (%s primitives), threads %-7s: build %.1f + compile %.1f = %.1f ms in the library (best of 5; %.1f ms with the harness: one object made "
              "and deleted per primitive, the quads copied to numpy), tree %s depth %s" % (which, n, threads, float(build), float(comp), float(build) + float(comp), float(ms), digest, depth))
    assert len({d for _, d in out.values()}) == 1, "trees differ"
    print("cpus %d; speed-up of the default over one thread: %.2fx" % (len(os.sched_getaffinity(0)), out["1"][0] / out["default"][0]))


if __name__ == "__main__":
    main()
