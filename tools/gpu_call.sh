#!/bin/bash
# scratch script of the CURRENT gpurun call (rewritten per call)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for rep in 1 2; do timeout 900 bash tools/variants.sh run planes pairs2 pairs0; done
tail -5 gpurun_out/variants/*.err
