#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
cp gpurun_out/accuracy_tests.json gpurun_out/accuracy_fingerprint.json $OUT/
