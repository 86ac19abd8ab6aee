#!/usr/bin/env python3
"""Per-kernel table of the TIMED steps of a tools/profile_config.sh directory (rocprofv3 rocpd databases).

The timed region of `bench.py --profile` is the END of the run (nothing follows it but the JSON line), so for every kernel
name the last (dispatches per step) x steps dispatches are the timed ones; dispatches per step = (dispatches of that name
in the run) // (passes of the step function in the run: setup + warm-up + timed, read from the bench line's own fields
or given).  Columns:
  calls/step, avg us per launch, us per step, share of the kernel time of a step        (stats pass)
  HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB -- FETCH_SIZE doubled per the gfx950 note of
     MI355X_MICROARCH.md (HBM section), each counter from its own pass -- and the GB/s / fraction of 8 TB/s they imply
  wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, mfma = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs) (x the CU count
     the counter aggregates over: reported raw), LDS bank conflict cycles, instruction mix per launch
usage: tools/profile_config_summary.py gpurun_out/prof_<tag> <timed steps> [passes]"""
import glob
import json
import os
import re
import sqlite3
import sys

src = sys.argv[1]
STEPS = int(sys.argv[2])
bench = None
try:
    bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
SETUP = {"cfg2": 25}.get((bench or {}).get("config", {}).get("workload", "cfg4")[:4], 5)
PASSES = int(sys.argv[3]) if len(sys.argv) > 3 else SETUP + (bench["warmup"] if bench else 3) + STEPS


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("(anonymous namespace)::", "").replace("kgcn::", "").replace("at::native::", "at::")
    n = re.sub(r"\(.*$", "", n)
    return n[:64]


def load(db):
    """-> {kernel: [(start, duration us)]} and the per-dispatch PMC rows {(kernel, counter): [(start, value)]} of one pass"""
    cur = sqlite3.connect(db).cursor()
    per, rows = {}, {}
    for name, start, d, vg, ag, lds in cur.execute("select name, start, duration, vgpr_count, accum_vgpr_count, lds_size "
                                                    "from kernels order by start"):
        per.setdefault(short(name), []).append((start, d / 1e3))
        meta[short(name)] = (vg, ag, lds)
    try:
        for k, c, start, v in cur.execute("select kernel_name, counter_name, start, value from counters_collection order by start"):
            rows.setdefault((short(k), c), []).append((start, v))
    except sqlite3.OperationalError:
        pass
    return per, rows


def timed_start(per):
    """Start time of the TIMED region of a pass.  The timed steps are the end of the run; a kernel of the step itself is launched by
    EVERY pass of the step function (set-up + warm-up + timed), i.e. (per step) x PASSES times, and its last (per step) x STEPS
    dispatches are the timed ones: the region starts with the earliest of those over all such kernels.  Kernels with fewer
    dispatches than that -- model construction, data generation, graph capture: the torch fill / copy kernels an earlier form of
    this script spread over the steps as "1 per step" -- do not define the region, and what they launched before it is not counted."""
    cands, names, t_end = [], [], max(r[-1][0] for r in per.values())
    for name, rows in per.items():
        n = round(len(rows) / PASSES)
        if n >= 1 and len(rows) >= n * (PASSES - 2):
            cands.append(rows[-min(len(rows), n * STEPS)][0])
            names.append(name)
    if not cands:                                            # (no kernel ran once per pass: count everything)
        return min(r[0][0] for r in per.values())
    # a name that ALSO ran during set-up (torch copies: 90 uploads + one per step) passes the count test by accident and would pull the
    # region back by whole steps: the candidates of the step's own kernels all lie inside the first timed step, i.e. within one step
    # of the latest candidate
    latest = max(cands)
    step = (t_end - latest) / max(STEPS - 1, 1)
    keep = [(c, n) for c, n in zip(cands, names) if c >= latest - 1.25 * step]
    t0 = min(c for c, _ in keep)
    # ... and what was launched between the END of the previous step's last kernel and t0 (an index upload in front of the step's
    # first kernel) belongs to the first timed step
    prev_end = max([s_ + d * 1e3 for _, n in keep for s_, d in per[n] if s_ < t0] or [t0 - 1])
    return min(t0, prev_end + 1)


dur, meta, pmc = {}, {}, {}
for db in sorted(glob.glob(os.path.join(src, "*", "*.db"))):
    sub = os.path.basename(os.path.dirname(db))
    per, rows = load(db)
    if not per:
        continue
    t0 = timed_start(per)
    dur[sub] = {}
    for k, v in per.items():
        t = [d for s_, d in v if s_ >= t0]
        if t:
            dur[sub][k] = (len(t) / STEPS, t)
    if sub != "stats":
        for (k, c), vs in rows.items():
            t = [v for s_, v in vs if s_ >= t0]
            if t:
                pmc.setdefault(k, {})[c] = sum(t) / len(t)

avg = lambda xs: sum(xs) / len(xs)
st = dur.get("stats", {})
tot = sum(ps * avg(v) for ps, v in st.values()) or 1.0
print("== %s" % (bench["config"]["workload"] if bench else src))
if bench:
    print("== un-profiled bench line: %.4g %s, %.4f ms/step (%d steps)" % (bench["value"], bench["unit"], bench["ms_per_step"],
                                                                         bench["steps"]))
print("== rocprofv3 --kernel-trace --stats: kernels of one TIMED step (averages over %d steps); kernel time per step %.1f us"
      % (STEPS, tot))
if bench:
    ratio = tot / (bench["ms_per_step"] * 1e3)
    print("== kernel sum / un-profiled step = %.3f%s" % (ratio, "   ** > 1.02: the profiled passes ran slower than the bench run (clocks under the "
          "profiler): per-kernel us here are UPPER bounds of the un-profiled step's **" if ratio > 1.02 else ""))
print("%-64s %5s %9s %9s %6s | %10s %8s %6s | %6s %6s %9s | %9s %8s %8s %8s %8s" % (
    "kernel", "n/stp", "avg us", "us/step", "share", "HBM MB", "GB/s", "frac", "wait", "mfma", "ldsconf", "valu", "mfma_i",
    "lds_i", "vmem_rd", "vmem_wr"))
for k, (ps, v) in sorted(st.items(), key=lambda kv: -kv[1][0] * avg(kv[1][1])):
    a = avg(v)
    if ps * a < 0.002 * tot:
        continue
    c = pmc.get(k, {})
    line = "%-64s %5s %9.1f %9.1f %5.1f%% |" % (k, ("%d" % round(ps)) if abs(ps - round(ps)) < 0.02 else ("%.2f" % ps), a, ps * a, 100 * ps * a / tot)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        line += " %10.2f %8.0f %6.3f |" % (b / 1e6, b / a / 1e3, b / a / 1e3 / 8000.0)
    else:
        line += " %10s %8s %6s |" % ("-", "-", "-")
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        line += " %6.3f %6.3f %9.3g |" % (c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                                          c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(c.get("SQ_BUSY_CYCLES", 1), 1) / 4,
                                          c.get("SQ_LDS_BANK_CONFLICT", 0))
    else:
        line += " %6s %6s %9s |" % ("-", "-", "-")
    line += " %9.3g %8.3g %8.3g %8.3g %8.3g" % tuple(c.get(n, float("nan")) for n in (
        "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    print(line)
print("== registers / LDS of those kernels")
for k, (ps, v) in sorted(st.items(), key=lambda kv: -kv[1][0] * avg(kv[1][1])):
    if ps * avg(v) >= 0.01 * tot:
        print("%-64s vgpr %s agpr %s lds %s" % ((k,) + meta[k]))
if "pmc_grbm" in dur:
    print("== effective clock in the GRBM pass: GRBM_GUI_ACTIVE / 8 XCDs / duration")
    for k, (ps, v) in sorted(dur["pmc_grbm"].items(), key=lambda kv: -kv[1][0] * avg(kv[1][1]))[:8]:
        c = pmc.get(k, {})
        if "GRBM_GUI_ACTIVE" in c:
            print("%-64s %.3f GHz (%.1f us in that pass, %.1f us in the stats pass)" % (
                k, c["GRBM_GUI_ACTIVE"] / 8 / avg(v) / 1e3, avg(v), avg(st[k][1]) if k in st else float("nan")))

# HBM bytes per launch of every kernel with both PMC passes -> <dir>/traffic.json (copied to profiles/traffic_<cfg>.json; bench.py
# fills `roofline.traffic` from it and flags it stale when the kernel sources changed since)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import source_hash
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tj = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (tools/profile_config.sh): FETCH_SIZE and WRITE_SIZE collected in "
                  "separate runs, KiB units, FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (HBM section); averages over "
                  "the dispatches of the timed steps",
      "workload": bench["config"]["workload"] if bench else None, "kernels": {}, "kernel_sources_sha256": source_hash.sources_sha256()}
for k, (ps, v) in st.items():
    c = pmc.get(k, {})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        tj["kernels"][k] = {"bytes": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), "fetch_size_kib": c["FETCH_SIZE"],
                            "write_size_kib": c["WRITE_SIZE"], "avg_us": round(avg(v), 2), "launches_per_step": ps,
                            "us_per_step": round(ps * avg(v), 2)}
json.dump(tj, open(os.path.join(src, "traffic.json"), "w"), indent=1)
