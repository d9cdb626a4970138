#!/usr/bin/env python3
"""Same-box A/B of GPUART_HIP_* environment settings: tools/run_passes.py for every setting in turn, several rounds; best and median
ms/pass.   python3 tools/knob_ab.py [-k PASSES] [-r ROUNDS] "NAME=V ..." "NAME=V ..."   ("" = defaults; WORKLOAD=... is passed on)"""
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
ap.add_argument("settings", nargs="+")
a = ap.parse_args()
res = {s: [] for s in a.settings}
for _ in range(a.r):
    for s in a.settings:
        env = dict(os.environ)
        for kv in s.split():
            k, v = kv.split("=")
            env[k if k in ("WORKLOAD", "GPUART_MODE", "GPUART_LIBDIR") else "GPUART_HIP_" + k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_passes.py"), str(a.k), "3"], env=env, capture_output=True, text=True, check=True).stdout
        res[s].append(min(float(x) for x in re.findall(r"([0-9.]+) ms/pass", out)))
for s in a.settings:
    print("%-40s best %.3f  median %.3f ms/pass  %s" % (s or "(defaults)", min(res[s]), statistics.median(res[s]), " ".join("%.3f" % x for x in res[s])), flush=True)
