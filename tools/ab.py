#!/usr/bin/env python3
"""Same-box A/B of library builds (tools/ab_build.sh): runs tools/run_passes.py for every variant in turn, several
rounds, and prints each variant's best and median ms/pass.   python3 tools/ab.py [-k PASSES] [-r ROUNDS] name ..."""
import argparse
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("-k", type=int, default=64)
ap.add_argument("-r", type=int, default=4)
ap.add_argument("names", nargs="+")
a = ap.parse_args()
res = {n: [] for n in a.names}
frames = {n: set() for n in a.names}
for _ in range(a.r):
    for n in a.names:
        env = dict(os.environ)
        if n != "default":
            env["GPUART_LIBDIR"] = os.path.join(ROOT, "gpuart_amd", "lib_ab", n)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_passes.py"), str(a.k), "3"], env=env,
                             capture_output=True, text=True, check=True).stdout
        res[n].append(min(float(x) for x in re.findall(r"([0-9.]+) ms/pass", out)))
        frames[n].update(re.findall(r"frame ([0-9a-f]+)", out))
print("frames: %s" % ("identical in every variant and round (%s)" % next(iter(frames[a.names[0]])) if len(set().union(*frames.values())) == 1
                      else "DIFFER: %s" % {n: sorted(f) for n, f in frames.items()}), flush=True)
for n in a.names:
    print("%-24s best %.3f  median %.3f ms/pass  %s" % (n, min(res[n]), statistics.median(res[n]), " ".join("%.3f" % x for x in res[n])), flush=True)
