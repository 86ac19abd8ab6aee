"""tools/profile_config_summary.py takes the TIMED region of a profiled bench run by time (DESIGN lesson 45): set-up kernels --
torch fills and uploads of model construction, the graph capture's warm-up -- must not be spread over the steps, and a kernel that
runs in front of the step's first own kernel (the pinned index upload) belongs to the step.  A synthetic rocpd database stands
in for a rocprofv3 pass."""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timed_region_is_taken_by_time(tmp_path):
    d = tmp_path / "prof"
    (d / "stats").mkdir(parents=True)
    con = sqlite3.connect(str(d / "stats" / "x.db"))
    cur = con.cursor()
    cur.execute("create table kernels(name, start, duration, vgpr_count, accum_vgpr_count, lds_size)")
    t = 0
    for _ in range(90):                                   # uploads of data generation: as many as 3 per pass of the step function
        cur.execute("insert into kernels values(?,?,?,?,?,?)", ("__amd_rocclr_copyBuffer", t, 3000, 8, 0, 0)); t += 5000
    for _ in range(31):                                   # parameter initialisation
        cur.execute("insert into kernels values(?,?,?,?,?,?)", ("at::FillFunctor", t, 2500, 8, 0, 0)); t += 5000
    step = (("__amd_rocclr_copyBuffer", 3000), ("assemble", 4000), ("fwd", 24000), ("small", 4500), ("small", 4500), ("bwd", 40000), ("adam", 5000))
    for _ in range(30):                                   # 5 set-up + 5 warm-up + 20 timed passes
        t += 20000
        for name, dur in step:
            cur.execute("insert into kernels values(?,?,?,?,?,?)", (name, t, dur, 8, 0, 0)); t += dur + 500
    con.commit(); con.close()
    json.dump({"config": {"workload": "cfg1: synthetic"}, "warmup": 5, "steps": 20, "value": 1.0, "unit": "graphs/sec", "ms_per_step": 0.1},
              open(str(d / "bench.json"), "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_config_summary.py"), str(d), "20"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    rows = {}
    for line in r.stdout.splitlines():
        f = line.split()
        if len(f) > 4 and f[0] in ("fwd", "bwd", "adam", "small", "assemble", "__amd_rocclr_copyBuffer", "at::FillFunctor"):
            rows.setdefault(f[0], (f[1], float(f[2])))          # (the per-kernel table comes first; the register table repeats the names)
    assert "at::FillFunctor" not in rows, "a set-up kernel was counted into the step"
    assert rows["fwd"] == ("1", 24.0) and rows["bwd"] == ("1", 40.0) and rows["adam"][0] == "1" and rows["assemble"][0] == "1"
    assert rows["small"][0] == "2"
    assert rows["__amd_rocclr_copyBuffer"][0] == "1", "the upload in front of the step's first kernel belongs to the step (and only that one)"
