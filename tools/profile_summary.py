#!/usr/bin/env python3
"""Summary of a tools/profile_round.sh directory.  Everything is computed over the TIMED dispatches of each profiled
bench run (the last --steps launches of a kernel: the first launches run while the chip's clocks and power state still
move -- 500 -> 720 -> 515 us on the fused backward -- and do not belong to the figure the bench line reports):
  * rocprofv3 --kernel-trace --stats durations (average, min, max) per hot-path kernel,
  * PMC per-dispatch averages, the effective clock GRBM_GUI_ACTIVE / 8 XCDs / duration of the same dispatches,
  * HBM traffic per launch (FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE as reported, KiB)
    -> traffic_cfg2.json,
  * agreement of the profiler's durations with the HIP-event times of the un-profiled bench line.
usage: tools/profile_summary.py gpurun_out/prof_<tag> [timed steps of the profiled runs = 30]"""
import glob
import json
import os
import re
import sqlite3
import sys

src = sys.argv[1]
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
short = lambda n: re.sub(r"\(.*$", "", n).replace("kgcn::", "").replace("void ", "")[:44]
HOT = ("graphconv", "spmm_tile", "spmm_slices", "reduce_partials")
dur, pmc, meta = {}, {}, {}
for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
    sub = os.path.basename(os.path.dirname(db))
    cur = sqlite3.connect(db).cursor()
    per = {}
    for name, start, d, vg, ag, lds in cur.execute("select name, start, duration, vgpr_count, accum_vgpr_count, lds_size "
                                                    "from kernels order by start"):
        if any(h in name for h in HOT):
            per.setdefault(short(name), []).append(d / 1e3)
            meta[short(name)] = (vg, ag, lds)
    dur[sub] = {k: v[-STEPS:] for k, v in per.items()}
    if sub != "stats":
        rows = {}
        for k, c, start, v in cur.execute("select kernel_name, counter_name, start, value from counters_collection order by start"):
            if any(h in k for h in HOT):
                rows.setdefault((short(k), c), []).append(v)
        for (k, c), vs in rows.items():
            vs = vs[-STEPS:]
            pmc.setdefault(k, {})[c] = (sub, len(vs), sum(vs) / len(vs))
avg = lambda xs: sum(xs) / len(xs)
print("== rocprofv3 --kernel-trace --stats, timed dispatches only (us)")
for k, v in dur.get("stats", {}).items():
    print("%-44s n=%-3d avg=%-8.1f min=%-8.1f max=%-8.1f vgpr+agpr=%s+%s lds=%s" % ((k, len(v), avg(v), min(v), max(v)) + meta[k]))
print("== PMC per-dispatch averages over the timed dispatches")
for k, cs in pmc.items():
    for c, (sub, n, v) in sorted(cs.items()):
        print("%-44s %-26s n=%-4d avg=%.6g" % (k, c, n, v))
bench = None
try:
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
traffic = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (tools/profile_round.sh): FETCH_SIZE and WRITE_SIZE "
                       "collected in separate runs, KiB units, FETCH_SIZE doubled per the gfx950 note in "
                       "MI355X_MICROARCH.md (HBM section); averages over the timed dispatches.",
           "graphs_per_launch": bench["config"]["graphs_per_gpu"] if bench else None}
print("== derived")
for k, cs in pmc.items():
    line = "%-44s" % k
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"][2], cs["WRITE_SIZE"][2]
        b = int((2 * f + w) * 1024)
        traffic[k.split("<")[0]] = {"fetch_size_kib": f, "write_size_kib": w, "bytes": b}
        line += " HBM %.4f GB/launch (fetch x2 %.4f + write %.4f)" % (b / 1e9, 2 * f * 1024 / 1e9, w * 1024 / 1e9)
    if "GRBM_GUI_ACTIVE" in cs:
        sub = cs["GRBM_GUI_ACTIVE"][0]
        d = avg(dur[sub][k])
        line += "  clock %.3f GHz in the counter pass (%.1f us), stats pass %.1f us" % (
            cs["GRBM_GUI_ACTIVE"][2] / 8 / d / 1e3, d, avg(dur["stats"][k]) if k in dur.get("stats", {}) else float("nan"))
    print(line)
if bench:
    r = bench["roofline"]
    print("== bench line (un-profiled run): value %.4g %s, %.4f ms/step, roofline.frac %.4f" % (
        bench["value"], bench["unit"], bench["ms_per_step"], r["frac"]))
    st = dur.get("stats", {})
    for name, ms in (("graphconv_bwd_pairs_kernel", r["launch_ms"]), ("graphconv_bwd_planes_kernel", r["launch_ms"]),
                     ("graphconv_fwd_full_kernel", r["fwd_kernel"]["launch_ms"])):
        if name in st:
            extra = avg(st["reduce_partials_kernel"]) if name.startswith("graphconv_bwd") and "reduce_partials_kernel" in st else 0.0
            prof = avg(st[name]) + extra
            print("%-30s HIP events median %.1f us (p10 %.1f, p90 %.1f)  |  rocprofv3 avg %.1f us%s  -> %+.1f %%" % (
                name, ms["median_ms"] * 1e3, ms["p10_ms"] * 1e3, ms["p90_ms"] * 1e3, prof,
                " (+ reduce_partials, in the event bracket)" if extra else "", 100 * (prof / (ms["median_ms"] * 1e3) - 1)))
    sp = r.get("spmm_kernel")
    if sp:
        print("batched SpMM (spmm_slices_kernel): forward %.1f us = %.4f of HBM peak, adjoint %.1f us = %.4f (HIP events, median)" % (
            sp["forward"]["median_ms"] * 1e3, sp["forward"]["frac"], sp["adjoint"]["median_ms"] * 1e3, sp["adjoint"]["frac"]))
# the kernel sources these byte counts were measured on: bench.py flags the figures as stale when they change
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_hash
traffic["kernel_sources_sha256"] = source_hash.sources_sha256(source_hash.CFG2_FILES)
json.dump(traffic, open(os.path.join(src, "traffic_cfg2.json"), "w"), indent=1)
