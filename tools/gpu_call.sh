#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06e; mkdir -p $OUT
timeout 600 python tools/bconv_bench.py > $OUT/bconv_c6.jsonl 2> $OUT/bconv.err; cat $OUT/bconv_c6.jsonl; tail -5 $OUT/bconv.err
timeout 900 python -m pytest tests/test_gpu_step_no_aten.py tests/test_gpu_model.py -m gpu -q -x -k "no_aten or captured or deferred" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
