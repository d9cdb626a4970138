#!/usr/bin/env python3
"""HBM traffic per k_trace launch from two rocprofv3 --pmc runs of the same bench command (FETCH_SIZE alone,
WRITE_SIZE alone — each needs the TCC counter slots to itself), corrected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes for gfx950 (FETCH_SIZE doubled; both counters are in KB).
    tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command the runs profiled>" """
import csv
import glob
import json
import os
import sys


def per_launch(d, counter):
    tot, n = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            k = row["Kernel_Name"]
            # fast-mode launches only: k_trace<COUNT = false, TYPES>
            if "k_trace<" in k and k.split("k_trace<")[1].split(">")[0].split(",")[0].strip() == "false":
                key = "k_trace"
            elif "k_shade<false>" in k:
                key = "k_shade"
            else:
                continue
            tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
            n[key] = n.get(key, 0) + 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


f = per_launch(sys.argv[1], "FETCH_SIZE")
w = per_launch(sys.argv[2], "WRITE_SIZE")
out = {
    "FETCH_SIZE_KB_per_launch_k_trace": f["k_trace"][0], "launches_k_trace": f["k_trace"][1],
    "WRITE_SIZE_KB_per_launch_k_trace": w["k_trace"][0],
    "FETCH_SIZE_KB_per_launch_k_shade": f.get("k_shade", (None, 0))[0],
    "WRITE_SIZE_KB_per_launch_k_shade": w.get("k_shade", (None, 0))[0],
    "hbm_bytes_per_launch": (2.0 * f["k_trace"][0] + w["k_trace"][0]) * 1024.0,
    "note": "k_trace (fast mode), average over its launches of `%s`; separate --pmc passes for FETCH_SIZE and WRITE_SIZE; "
            "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibrated there for wide coalesced streams; "
            "our 16-B/lane gathers are uncalibrated: read it as an upper estimate); KB -> bytes. The BVH (8.7 MB) is "
            "cache-resident (L2 / Infinity Cache), so HBM traffic is far below the algorithmic bytes." % sys.argv[4],
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
