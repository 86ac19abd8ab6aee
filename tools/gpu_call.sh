#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bconv_loop.py -m gpu -q -x > $OUT/pytest_loop.log 2>&1; tail -15 $OUT/pytest_loop.log
timeout 900 python -m pytest tests -m gpu -q -x -k "bconv or split or gat or maxpool or deferred or GCN" > $OUT/pytest_rel.log 2>&1; tail -4 $OUT/pytest_rel.log
timeout 600 python tools/bconv_bench.py > $OUT/bconv_c6.jsonl 2> $OUT/bconv.err; python - <<'PY'
import json
for l in open('gpurun_out/r06g/bconv_c6.jsonl'):
    d=json.loads(l); print(d['shape'], d['layer_fwd_bwd_ms']['median'], d['graphs_per_s'], d['bconv_forward'], d['bconv_adjoint'])
PY
tail -3 $OUT/bconv.err
