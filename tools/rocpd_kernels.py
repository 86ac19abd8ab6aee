#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 rocpd databases (one directory per pass under <dir>): launches, average / min duration of the
LAST `n` dispatches of every kernel, and the average of every collected counter over the same dispatches.
usage: tools/rocpd_kernels.py <dir> [n] [name filter]"""
import glob
import os
import re
import sqlite3
import sys

src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flt = sys.argv[3] if len(sys.argv) > 3 else ""


def short(s):
    s = re.sub(r"^void ", "", s).replace("kgcn::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", s)[:70]


for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
    sub = os.path.basename(os.path.dirname(db))
    cur = sqlite3.connect(db).cursor()
    per = {}
    for name, d in cur.execute("select name, duration from kernels order by start"):
        per.setdefault(short(name), []).append(d / 1e3)
    print("== %s" % sub)
    try:
        rows = {}
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection order by start"):
            rows.setdefault(short(k), {}).setdefault(c, []).append(v)
    except sqlite3.OperationalError:
        rows = {}
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if flt and flt not in k:
            continue
        t = v[-n:]
        line = "%-70s n=%4d avg %8.1f us min %8.1f" % (k, len(v), sum(t) / len(t), min(t))
        for c, vs in sorted(rows.get(k, {}).items()):
            vs = vs[-n:]
            line += "  %s=%.4g" % (c, sum(vs) / len(vs))
        print(line)
