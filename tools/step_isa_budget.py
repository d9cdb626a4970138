#!/usr/bin/env python3
"""Static instruction budget of ONE wide box step of k_trace<false, 38> (the reference-order kernel of triangle meshes), by purpose, from a
`hipcc -S --cuda-device-only` dump:   python3 tools/step_isa_budget.py dump.s
The sections are found by what they contain, not by label numbers (those change with every build): the loop that holds the four loads
of a node record; inside it the first block with v_med3 / v_max3 chains (the quick answers), the flat_load block (withdrawn answers: six
face tests), the ds_write pair (push), the loop with the ds_read pair (pop) ... Prints one line per section."""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2] if len(sys.argv) > 2 else "k_traceILb0ELi38E"
start = [i for i, l in enumerate(src) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l][0]
end = [i for i in range(start, len(src)) if "s_endpgm" in src[i]][0]
L = src[start:end + 1]


def kind(line):
    l = line.strip()
    if not l or l.startswith((".", ";")) or l.split(";")[0].strip().endswith(":"):
        return None
    op = l.split()[0]
    if op.startswith("v_"): return "VALU"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")): return "VMEM"
    return "other"


def count(a, b):
    c = dict(VALU=0, SALU=0, branch=0, LDS=0, VMEM=0, waitcnt=0, other=0)
    for l in L[a:b]:
        k = kind(l)
        if k: c[k] += 1
    return c


# basic blocks: a new block at every label / %bb comment
heads = [i for i, l in enumerate(L) if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", l)] + [len(L)]
blocks = [(heads[k], heads[k + 1]) for k in range(len(heads) - 1)]


def text(b): return "\n".join(L[b[0]:b[1]])


# the wide step: the first block that loads a whole record (two dwordx4 + two dwordx3 from one base)
rec = next(k for k, b in enumerate(blocks) if len(re.findall(r"global_load_dwordx[34] ", text(b))) >= 4 and "offset:48" in text(b))
quick = next(k for k in range(rec, len(blocks)) if text(blocks[k]).count("v_med3_f32") >= 2 and text(blocks[k]).count("v_max3_f32") >= 2)
faces = next(k for k in range(quick, len(blocks)) if "flat_load_dwordx3" in text(blocks[k]) or text(blocks[k]).count("v_med3_f32") >= 12)
push = next(k for k in range(faces, len(blocks)) if "ds_write_b64" in text(blocks[k]) and "ds_write_b32" in text(blocks[k]) and "global_store" not in text(blocks[k]))
spill = next(k for k in range(faces, push + 1) if "global_store_dwordx4" in text(blocks[k]))
tail = next(k for k in range(push, len(blocks)) if "s_bcnt1_i32_b64" in text(blocks[k]))
leaf = next(k for k in range(tail, len(blocks)) if "v_rcp_f32" in text(blocks[k]))
pop = next(k for k in range(leaf, len(blocks)) if re.search(r"ds_read_b64|ds_read_b32", text(blocks[k])) and "v_max_f32" in text(blocks[k]))
rows = [
    ("round head: which lanes descend; fetch of the 64-byte record", blocks[rec - 1][0], blocks[rec][1]),
    ("two quick box answers (box_quick.h)", blocks[quick][0], blocks[quick][1]),
    ("  [rare] a withdrawn answer: the six face tests, both boxes", blocks[faces][0], blocks[faces][1]),
    ("push: decision + write of the entry to the LDS ring", blocks[faces + 1][0], blocks[spill][0]),
    ("  [rare] ring full: the oldest entry goes to global memory", blocks[spill][0], blocks[spill][1]),
    ("push: the two LDS writes", blocks[push][0], blocks[push][1]),
    ("enter the lower child / go and pop: decision, new state from the ref", blocks[push + 1][0], blocks[tail - 1][1] if tail - 1 > push else blocks[push + 1][1]),
    ("round tail: ballots, leaf quorum, refill and thin-mode exits", blocks[tail - 1][0], blocks[leaf - 1][1]),
    ("pop loop, one trip: stack empty? entry spilled? read pe / he / ref, test", blocks[pop][0], blocks[pop + 2][1]),
]
print("k_trace<false, 38>: %s" % count(0, len(L)))
for name, a, b in rows:
    c = count(a, b)
    print("%-72s %4d VALU %4d SALU %3d branch %2d LDS %2d VMEM %2d waitcnt" % (name, c["VALU"], c["SALU"], c["branch"], c["LDS"], c["VMEM"], c["waitcnt"]))
