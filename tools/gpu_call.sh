#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "non_finite or repeats_bit" 2>&1 | tail -4
