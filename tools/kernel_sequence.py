#!/usr/bin/env python3
"""The kernels of the LAST timed step of a bench.py configuration in launch order (GPU box: runs rocprofv3 --kernel-trace itself).
usage: python tools/kernel_sequence.py --config cfg3 [bench args]"""
import glob, os, re, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = tempfile.mkdtemp(prefix="kseq_", dir="/tmp")
env = dict(os.environ, TMPDIR="/tmp")
steps = 4
subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "b", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--profile",
                "--steps", str(steps), "--warmup", "2"] + sys.argv[1:], cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
               stderr=subprocess.DEVNULL, check=False)
db = sorted(glob.glob(os.path.join(out, "**", "*.db"), recursive=True))[-1]
rows = list(sqlite3.connect(db).cursor().execute("select name, start, end from kernels order by start"))
names = [re.sub(r"\(.*$", "", re.sub(r"^void ", "", n)).replace("kgcn::", "")[:90] for n, _, _ in rows]
# the step = the period of the sequence at the tail
per = None
for p in range(5, len(names) // 2):
    if names[-p:] == names[-2 * p:-p]:
        per = p
        break
if per is None:
    raise SystemExit("no periodic tail found")
t0 = rows[-per][1]
prev_end = None
for (n, s, e), nm in zip(rows[-per:], names[-per:]):
    gap = "" if prev_end is None else "gap %5.1f" % ((s - prev_end) / 1e3)
    print("%8.1f us  %6.1f us  %-10s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, nm))
    prev_end = e
print("kernels per step: %d, span %.1f us" % (per, (rows[-1][2] - t0) / 1e3))
