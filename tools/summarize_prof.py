#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats, per-kernel PMC averages) into small text files.
usage: tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<name>.txt"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(src, dst):
    lines = []
    for f in sorted(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        lines.append("== kernel stats (rocprofv3 --kernel-trace --stats): %s" % os.path.relpath(f, src))
        rows = list(csv.DictReader(open(f)))
        lines.append("%-64s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows:
            lines.append("%-64s %8s %14s %12.0f %8s" % (r["Name"][:64], r["Calls"], r["TotalDurationNs"],
                                                        float(r["AverageNs"]), r["Percentage"]))
    for sub in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for f in sorted(glob.glob(os.path.join(sub, "**", "*counter_collection.csv"), recursive=True)):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            lines.append("== PMC per-dispatch averages: %s" % os.path.relpath(f, src))
            for k in sorted(acc):
                for c in sorted(acc[k]):
                    v = acc[k][c]
                    lines.append("%-64s %-28s n=%-5d avg=%.6g" % (k[:64], c, len(v), sum(v) / len(v)))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
