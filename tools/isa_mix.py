#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S dump:  tools/isa_mix.py file.s kernel_substring"""
import collections
import sys

s = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = [i for i, l in enumerate(s) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = [i for i in range(start, len(s)) if "s_endpgm" in s[i]][0]
c = collections.Counter()
for l in s[start + 1:end]:
    l = l.strip()
    if not l or l.startswith((".", ";")):
        continue
    if l.split(";")[0].strip().endswith(":"):
        c["label"] += 1
        continue
    op = l.split()[0]
    if op.startswith("v_"): c["VALU"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): c["BRANCH"] += 1
    elif op.startswith("s_waitcnt"): c["WAIT"] += 1
    elif op.startswith("s_"): c["SALU"] += 1
    elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): c["VMEM"] += 1
    elif op.startswith("ds_"): c["LDS"] += 1
    else: c[op] += 1
print(dict(c))
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(s[start:end + 1]))
