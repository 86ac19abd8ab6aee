#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) output: kernel stats and per-kernel PMC averages.
usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>.txt"""
import glob
import os
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*$", "", name)[:48]


def main(src, dst):
    lines = []
    for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
        sub = os.path.basename(os.path.dirname(db))
        cur = sqlite3.connect(db).cursor()
        if sub.startswith("stats"):
            lines.append("== %s: rocprofv3 --kernel-trace --stats (durations in us)" % sub)
            lines.append("%-48s %6s %12s %10s %7s %5s %5s" % ("kernel", "calls", "total_us", "avg_us", "pct", "vgpr", "lds"))
            meta = {r[0]: r[1:] for r in cur.execute(
                "select name, max(vgpr_count)+max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by 1")}
            for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
                m = meta.get(name, (0, 0, 0, 0))
                lines.append("%-48s %6d %12.1f %10.1f %7.2f %5s %5s" % (short(name), calls, total, avg, pct, m[0], m[1]))
        else:
            lines.append("== %s: PMC per-dispatch averages" % sub)
            q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                 "group by 1,2 order by 1,2")
            for k, c, n, v in cur.execute(q):
                lines.append("%-48s %-28s n=%-4d avg=%.6g" % (short(k), c, n, v))
    txt = "\n".join(lines) + "\n"
    open(dst, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
