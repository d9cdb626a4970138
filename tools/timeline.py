#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace csv: busy union, per-kernel totals, and the last N launches in time
order.   tools/timeline.py <x_kernel_trace.csv> [last_n]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sel = rows[-last:]
t0 = sel[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, n, q in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for _, e, _, _ in sel) - t0
print("last %d launches: span %.1f us, busy (union) %.1f us, sum of durations %.1f us" % (len(sel), span / 1e3, busy / 1e3, sum(e - s for s, e, _, _ in sel) / 1e3))
for s, e, n, q in sel:
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:50]))
