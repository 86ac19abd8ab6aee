#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
timeout 600 bash tools/variants.sh run psync 2>&1 | cut -c1-500
for rep in 1 2; do VB_TIMING_ONLY=1 timeout 900 bash tools/variants.sh run head psync 2>&1 | cut -c1-1500; done
