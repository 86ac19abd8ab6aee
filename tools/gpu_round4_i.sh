#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python tools/gemmh_bench.py --rows 117888 --shapes 256x256 2>&1 >/dev/null | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
