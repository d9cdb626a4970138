set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/tcp_ubench; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $OUT -o x -- $R/tools/ubench/ubench > $OUT/log.txt 2>&1
python3 - <<PY
import csv, collections
dur = collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/x_kernel_trace.csv")):
    dur[(r["Kernel_Name"], r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(dict)
for r in csv.DictReader(open("$OUT/x_counter_collection.csv")):
    acc[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
seen = collections.Counter()
for (k, d), c in acc.items():
    if not ("loadcost" in k or "width" in k or "gather" in k or "k_node" in k): continue
    seen[k] += 1
    if seen[k] % 2 == 1: continue   # every measurement launches twice (warm-up, timed): keep the second
    t = dur[(k, d)] * 1e-9
    grid = "?"
    print("%-44s %7.3f ms  TOTAL_ACCESSES %.3e (%.3e /s)  CACHE_ACCESSES %.3e (%.3e /s)  TCC_READ_REQ %.3e" % (
        k.split("(")[0][-42:], t * 1e3, c.get("TCP_TOTAL_ACCESSES_sum", 0), c.get("TCP_TOTAL_ACCESSES_sum", 0) / t,
        c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / t, c.get("TCP_TCC_READ_REQ_sum", 0)))
PY
