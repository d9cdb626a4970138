#!/usr/bin/env python3
"""Cross-check of bench.py's HIP-event timing against rocprofv3: the average duration of the fast-mode k_trace launches
of the timed region (the last `launches` ones of the kernel trace) beside `roofline.kernel_avg_ms` of the bench line the
same process printed.   tools/trace_agreement.py <x_kernel_trace.csv> <bench stdout of that run>"""
import csv
import json
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_trace<" in k:
        args = [a.strip() for a in k.split("k_trace<")[1].split(">")[0].split(",")]
        # the default walk's variants carry GD_REF_ORDER (0x20) in their type mask; the opt-in nearest-first section of bench.py
        # (`nearest_first_opt_in`, after the timed region) launches the variants without it
        if args[0] == "false" and (int(args[1]) & 0x20):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
line = [l for l in open(sys.argv[2]).read().split("\n") if l.startswith("{")][-1]
d = json.loads(line)
n = d["roofline"]["kernel_launches"]
# after the timed repetitions bench.py renders ONE more K-pass sequence with the default walk (the frame the opt-in walk's is compared
# with): its launches are the last n / repeats of the trace; the n before them are the timed ones
tail = n // max(1, d["repeats"]) if d.get("nearest_first_opt_in") else 0
sel = rows[-(n + tail):len(rows) - tail]
assert len(sel) == n, (len(sel), n, len(rows))
avg = sum(e - s for s, e in sel) / len(sel) / 1e6
print("command: python3 bench.py --steps %d --warmup %d --no-cpu-baseline --no-profile (under rocprofv3 --kernel-trace --stats)" % (d["steps"], d["warmup"]))
print("k_trace launches in the timed region: %d" % n)
print("rocprofv3 kernel trace, average duration of those launches: %.4f ms" % avg)
print("bench.py HIP events (roofline.kernel_avg_ms):                %.4f ms" % d["roofline"]["kernel_avg_ms"])
print("ratio: %.3f" % (avg / d["roofline"]["kernel_avg_ms"]))
print("bench line of the profiled run: value %.1f %s, ms_per_step %.4f, k_trace launch durations summed per pass %.3f ms, concurrency %.2f"
      % (d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms_summed_per_pass"], d["roofline"]["kernel_concurrency"]))
