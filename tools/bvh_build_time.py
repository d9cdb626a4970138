#!/usr/bin/env python3
"""Times the host BVH build on Scene D: sequential (GPUART_BVH_FORK_LEVELS=0, the reference's
algorithm as it stands) against the task-parallel build that is the default, in child processes so that the
environment variable is read afresh; both must give the same bytes."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, hashlib
sys.path.insert(0, %r)
from gpuart_amd import binding as B, synth_scenes as S
p = B.make_prims(S.scene_d())
best = 1e9
for i in range(5):
    t = time.perf_counter(); q, depth = B.compile_bvh(p); best = min(best, time.perf_counter() - t)
print("%%.1f %%s %%d" %% (best * 1e3, hashlib.sha256(q.tobytes()).hexdigest()[:16], depth))
""" % ROOT


def main():
    out = {}
    for levels in ("0", "2", "4", "6"):
        env = dict(os.environ, GPUART_BVH_FORK_LEVELS=levels)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, check=True)
        ms, digest, depth = r.stdout.split()
        out[levels] = (float(ms), digest)
        print("fork levels %s: build+compile %.1f ms (best of 5), tree %s depth %s" % (levels, float(ms), digest, depth))
    assert len({d for _, d in out.values()}) == 1, "trees differ"
    print("cpus %d; speed-up of the default (4 levels) over sequential: %.2fx"
          % (len(os.sched_getaffinity(0)), out["0"][0] / out["4"][0]))


if __name__ == "__main__":
    main()
