#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc run:  tools/pmc_summary.py <dir>  (reads every *counter_collection.csv below it)."""
import collections
import csv
import glob
import os
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[name][row["Counter_Name"]] += 1
for name in sorted(acc):
    c = acc[name]
    n = max(cnt[name].values())
    line = "%-46s launches %5d" % (name[:46], n)
    for k in sorted(c):
        line += "  %s %.4g" % (k, c[k] / cnt[name][k])
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and c["SQ_ACTIVE_INST_VALU"]:
        line += "  | lane_util %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]))
    print(line)
