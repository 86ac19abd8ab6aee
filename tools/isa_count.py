#!/usr/bin/env python3
"""Instruction mix of one kernel's loops in a hipcc -S dump (development tool).
usage: isa_count.py file.s kernel_substring"""
import re
import sys
from collections import Counter

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]


def cat(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): return "vmem_rd"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")): return "vmem_wr"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


# basic blocks
blocks, cur, name = [], [], "entry"
for l in body:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith((";", ".", "//")): continue
    op = t.split()[0]
    cur.append((op, t))
blocks.append((name, cur))
tot = Counter()
for name, ins in blocks:
    c = Counter(cat(op) for op, _ in ins)
    tot.update(c)
    if len(ins) >= 40:
        tgt = [t for op, t in ins if op.startswith(("s_cbranch", "s_branch"))]
        print("%-12s %5d  %s   -> %s" % (name, len(ins), dict(c), [t.split()[-1] for t in tgt]))
print("total", sum(tot.values()), dict(tot))
if len(sys.argv) > 3:
    want = sys.argv[3]
    for name, ins in blocks:
        if name == want:
            c = Counter(op for op, _ in ins)
            for op, n in c.most_common(60): print("   %-40s %d" % (op, n))
