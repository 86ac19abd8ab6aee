#!/usr/bin/env python3
"""gpurun_out/ of one tools/evidence_round.sh call -> profiles/<tag>_* (the tracked, judged copies).  usage: collect_evidence.py <tag>"""
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def cp(src, dst):
    if os.path.exists(src):
        shutil.copyfile(src, os.path.join(P, dst))
        print("profiles/%s" % dst)
    else:
        print("missing: %s" % os.path.relpath(src, ROOT))


cp(os.path.join(G, "prof_%s" % tag, "summary.txt"), "%s_cfg2_fused.txt" % tag)
cp(os.path.join(G, "prof_%s" % tag, "bench.json"), "%s_cfg2_bench.json" % tag)
for c in ("cfg1", "cfg3", "cfg4", "cfg5"):
    cp(os.path.join(G, "prof_%s_%s" % (tag, c), "summary.txt"), "%s_%s_rocprof.txt" % (tag, c))
    cp(os.path.join(G, "prof_%s_%s" % (tag, c), "bench.json"), "%s_%s_bench.json" % (tag, c))
cp(os.path.join(G, tag, "lines.txt"), "%s_bench_lines.txt" % tag)
with open(os.path.join(P, "%s_companions_and_one_rank_rccl.jsonl" % tag), "w") as f:
    for n in ("cfg2_normalize", "cfg2_graphs4096", "cfg2_forcedist", "cfg4_forcedist", "cfg5_forcedist"):
        p = os.path.join(G, tag, n + ".json")
        if os.path.exists(p):
            f.write(open(p).read().strip().split("\n")[-1] + "\n")
for n, d in (("backward_forms.txt", "backward_forms_same_box.txt"), ("pairs_probe.txt", "bwd_pairs_probe.txt"), ("skeleton.txt", "microbench_bwd_memory_skeleton.txt"),
             ("bconv_c6.jsonl", "bconv_c6.jsonl"), ("fuzz_gpu.txt", "fuzz_gpu.txt"), ("fuzz_gemmh.txt", "fuzz_gemmh.txt"), ("pytest.log", "pytest_gpu.log")):
    cp(os.path.join(G, tag, n), "%s_%s" % (tag, d))
for c in ("cfg2", "cfg1", "cfg3", "cfg4", "cfg5"):
    print("traffic_%s.json:" % c, "present" if os.path.exists(os.path.join(P, "traffic_%s.json" % c)) else "MISSING")
