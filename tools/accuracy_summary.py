#!/usr/bin/env python3
"""gpurun_out/accuracy_tests.json (written by tests/conftest.py at the end of a `pytest -m gpu` session: every close() call's measured
maximum error, tolerance and reference magnitude per test) -> a compact copy for profiles/: floats to three digits, checks of a test
that share a label merged (worst error kept), plus the ten checks closest to their tolerance.
usage: python tools/accuracy_summary.py gpurun_out/accuracy_tests.json profiles/r04_accuracy.json"""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
d = json.load(open(src))
f3 = lambda v: float("%.3g" % v)
out, flat = {}, []
for test, checks in sorted(d.items()):
    row = {}
    for what, c in checks.items():
        row[what] = {"err": f3(c["max_abs_err"]), "tol": f3(c["tolerance"]), "ref_max": f3(c["ref_max_abs"]), "calls": c["calls"]}
        if c["tolerance"] > 0:
            flat.append((c["max_abs_err"] / c["tolerance"], test, what))
    out[test] = row
flat.sort(reverse=True)
doc = {"_comment": "measured max |HIP - reference| of every tests/ close() call in one `pytest -m gpu` session on MI355X (err), its "
                   "tolerance (tol), the largest reference magnitude (ref_max); tests/conftest.py records, tools/accuracy_summary.py condenses",
       "tests": len(out), "checks": sum(len(v) for v in out.values()),
       "closest_to_tolerance": [{"err_over_tol": f3(r), "test": t, "check": w} for r, t, w in flat[:10]],
       "by_test": out}
json.dump(doc, open(dst, "w"), indent=0, separators=(",", ":"))
print("%d tests, %d checks -> %s" % (doc["tests"], doc["checks"], dst))
