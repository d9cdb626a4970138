#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a `hipcc -S --cuda-device-only` dump.   python3 tools/kernel_resources.py dump.s [filter]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g(r"\.name")
    if flt in name:
        print("%-70s vgpr %s agpr %s sgpr %s spill %s scratch %s lds %s" % (name[:70], g(r"\.vgpr_count"), blk.split()[0], g(r"\.sgpr_count"),
              g(r"\.vgpr_spill_count"), g(r"\.private_segment_fixed_size"), g(r"\.group_segment_fixed_size")))
