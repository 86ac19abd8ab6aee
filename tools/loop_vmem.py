#!/usr/bin/env python3
"""VMEM instructions and vmcnt waits of a kernel's outermost hot loop, in program order (development tool).
usage: loop_vmem.py file.s kernel_substring"""
import re
import sys
txt = open(sys.argv[1]).read().split("\n")
s = next(i for i, l in enumerate(txt) if l.startswith("_Z") and sys.argv[2] in l and ":" in l)
e = next(i for i in range(s, len(txt)) if txt[i].startswith(".Lfunc_end"))
lines = txt[s:e]
hdr = [i for i, l in enumerate(lines) if "This Loop Header: Depth=1" in l]
start = hdr[-1]
lab = re.match(r"^(\.LBB\d+_\d+):", lines[start]).group(1)
end = max(i for i, l in enumerate(lines) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + "$", l.strip()))
n = 0
for i in range(start, end + 1):
    t = lines[i].strip()
    if not t or t[0] in ";.":
        continue
    op = t.split()[0]
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        n += 1
        print(i, n, t[:80])
    elif "vmcnt" in t:
        print(i, "    ", t)
tail = [lines[i].strip() for i in range(end - 30, end + 1) if "v_mov" in lines[i] or "accvgpr" in lines[i]]
print("moves before the back edge:", tail)
