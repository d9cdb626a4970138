#!/usr/bin/env python3
"""ms per pass of cfg3 for K passes cut into runs of exactly L passes (+ a remainder run), K and L swept: the data behind run_planner.h's
run length (profiles/r05/run_length_sweep.txt was taken on the library of the commit before the planner change, whose runs were exactly
MAX_BATCH passes + a remainder; today's planner evens the runs out: K = 20, L = 6 becomes 4 x 5).   python3 tools/run_length_sweep.py K0 K1 [L0 L1]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
l0, l1 = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2, 8)
print("K \\ L  " + " ".join("%6d" % l for l in range(l0, l1 + 1)) + "    runs at the best L", flush=True)
for k in range(k0, k1 + 1):
    row = []
    for l in range(l0, l1 + 1):
        env = dict(os.environ, GPUART_HIP_PLAN_RUN_PERCENT="400", GPUART_HIP_MAX_BATCH=str(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_passes.py"), str(k), "4"], env=env, capture_output=True, text=True, check=True).stdout
        row.append(min(float(x) for x in re.findall(r"([0-9.]+) ms/pass", out)))
    b = row.index(min(row)) + l0
    print("%-6d  " % k + " ".join("%6.3f" % v for v in row) + "    L=%d: %d runs" % (b, (k + b - 1) // b), flush=True)
