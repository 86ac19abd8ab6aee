#!/usr/bin/env python3
"""Per-kernel PMC averages + kernel-trace stats out of the rocprofv3 (rocpd sqlite) directories that
tools/pmc_variant.sh writes.  usage: tools/pmc_summary.py gpurun_out/pmc_<name> [kernel substring]"""
import glob
import os
import re
import sqlite3
import sys

src = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "graphconv"
short = lambda n: re.sub(r"\(.*$", "", n).replace("kgcn::", "")[:40]
for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
    sub = os.path.basename(os.path.dirname(db))
    cur = sqlite3.connect(db).cursor()
    if sub.startswith("stats"):
        print("== %s: rocprofv3 --kernel-trace --stats (us)" % sub)
        for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
            print("%-40s calls=%-5d total=%-12.1f avg=%-10.2f pct=%.2f" % (short(name), calls, total, avg, pct))
    else:
        print("== %s: PMC per-dispatch averages" % sub)
        for k, c, n, v in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                      "group by 1,2 order by 1,2"):
            if key in k:
                print("%-40s %-28s n=%-4d avg=%.6g" % (short(k), c, n, v))
