#!/usr/bin/env python3
"""How the runs of ONE pass sequence overlap: a rocprofv3 --kernel-trace csv of `tools/run_passes.py K REPS` -> for the last sequence,
per hardware queue when its kernels start and end and how busy it is, and kernels in flight per 250 us.
tools/run_shape_trace.py <x_kernel_trace.csv> <launches of the last sequence to look at>"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"[<(].*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
# the last sequence = everything after the last gap of more than 300 us without any kernel
ends = 0
cut = 0
for i, (s, e, n, q) in enumerate(rows):
    if i and s > ends + 300000:
        cut = i
    ends = max(ends, e)
sel = rows[cut:]
t0 = sel[0][0]
span = max(e for _, e, _, _ in sel) - t0
print("last sequence: %d launches, span %.1f us, sum of durations %.1f us (mean concurrency %.2f)" % (len(sel), span / 1e3, sum(e - s for s, e, _, _ in sel) / 1e3, sum(e - s for s, e, _, _ in sel) / span))
byq = collections.defaultdict(list)
for s, e, n, q in sel:
    byq[q].append((s, e, n))
for q, v in sorted(byq.items(), key=lambda kv: kv[1][0][0]):
    busy = sum(e - s for s, e, _ in v)
    kinds = collections.Counter(n for _, _, n in v)
    print("queue %-4s %3d launches  first start %8.1f  last end %8.1f  busy %8.1f us  %s" % (q, len(v), (v[0][0] - t0) / 1e3, (max(e for _, e, _ in v) - t0) / 1e3, busy / 1e3, dict(kinds)))
B = 250000
nb = span // B + 1
infl = [0.0] * nb
tr = [0.0] * nb
for s, e, n, q in sel:
    for b in range((s - t0) // B, (e - t0) // B + 1):
        lo, hi = max(s - t0, b * B), min(e - t0, (b + 1) * B)
        if hi > lo:
            infl[b] += (hi - lo) / B
            if n == "k_trace": tr[b] += (hi - lo) / B
print("kernels in flight per 250 us (all / k_trace):")
print(" ".join("%.1f/%.1f" % (a, b) for a, b in zip(infl, tr)))
