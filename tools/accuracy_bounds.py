#!/usr/bin/env python
"""tests/golden/accuracy_bounds.json + profiles/<tag>_accuracy.json from the accuracy record of a full `-m gpu` session
(gpurun_out/accuracy_tests.json, written by tests/conftest.py).

  relative bound of a comparison = max(10 x its largest measured err / max(1, |ref|), 2^-22)

tests/conftest.py::accuracy_bound applies it on top of the tolerance written in the test (the smaller one wins).
usage: python tools/accuracy_bounds.py gpurun_out/accuracy_tests.json r05     (run it on a record taken with KGCN_NO_RATCHET unset or
set: the bound only depends on the measured errors)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOOR = 2.0 ** -22


def main():
    rec = json.load(open(sys.argv[1]))
    tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
    bounds, checks, loose_before, loose_after, worst = {}, 0, 0, 0, []
    for test, labels in sorted(rec.items()):
        for what, e in sorted(labels.items()):
            mag1 = max(1.0, e["ref_max_abs"])
            rel = e.get("max_rel_err", e["max_abs_err"] / mag1)
            b = max(10.0 * rel, FLOOR)
            bounds.setdefault(test, {})[what] = b
            checks += 1
            written = e.get("written_tolerance") or e["tolerance"]
            err = max(e["max_abs_err"], 1e-300)
            new_tol = min(written, b * mag1)
            if written / err > 30 and written > FLOOR * mag1 * 1.0001:
                loose_before += 1
            if new_tol / err > 30 and new_tol > FLOOR * mag1 * 1.0001:
                loose_after += 1
            worst.append((new_tol / mag1, test, what, e["max_abs_err"], new_tol, e["ref_max_abs"]))
    worst.sort(reverse=True)
    fpath = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "accuracy_fingerprint.json")
    fp = json.load(open(fpath))["fingerprint"] if os.path.exists(fpath) else None     # tests/conftest.py applies the table only on a match
    out = {"made_from": os.path.basename(sys.argv[1]), "fingerprint": fp, "tests_with_comparisons": len(rec), "comparisons": checks,
           "rule": "relative bound = max(10 x max err / max(1, |ref|), 2^-22); close() uses min(written tolerance, bound x max(1, |ref|))",
           "bounds": bounds}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "accuracy_bounds.json"), "w"), indent=0, sort_keys=True)
    summary = {"tests_with_comparisons": len(rec), "comparisons": checks,
               "looser_than_30x_measured_before_the_ratchet": loose_before,
               "looser_than_30x_measured_with_the_ratchet (only comparisons sitting on the fp32 floor 2^-22 max(1,|ref|) remain above it)": loose_after,
               "largest_effective_tolerance_relative_to_max1_ref": [
                   {"tol_over_max1_ref": w[0], "test": w[1], "what": w[2], "max_abs_err": w[3], "tolerance": w[4], "ref_max_abs": w[5]}
                   for w in worst[:25]],
               "record": rec}
    json.dump(summary, open(os.path.join(ROOT, "profiles", "%s_accuracy.json" % tag), "w"), indent=1, sort_keys=True)
    print("comparisons %d in %d tests; tolerance / measured error > 30: %d before, %d with the ratchet" % (checks, len(rec), loose_before, loose_after))
    for w in worst[:8]:
        print("  tol/max(1,|ref|) %.2e  err %.2e  %s :: %s" % (w[0], w[3], w[1][-70:], w[2][:50]))


if __name__ == "__main__":
    main()
