#!/bin/bash
# round 5, call a: full GPU suite at HEAD (accuracy record of every close()), bench lines of every config on this box,
# SURVEY 8(d) companions of cfg2 and the one-rank RCCL lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest.log
cp gpurun_out/accuracy_tests.json gpurun_out/r05a/accuracy_tests.json
bash tools/bench_lines.sh r05a > gpurun_out/r05a/lines.txt 2>&1
python bench.py --normalize --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05a/cfg2_normalize.json
python bench.py --graphs 4096 --graph --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05a/cfg2_graphs4096.json
for c in cfg2 cfg4 cfg5; do python bench.py --config $c --force-dist --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r05a/${c}_forcedist.json; done
tail -3 gpurun_out/r05a/pytest.log; cat gpurun_out/r05a/lines.txt
