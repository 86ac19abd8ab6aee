#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
for c in cfg3 cfg4; do ATEN_WHO=1 timeout 300 python tools/aten_in_step.py --config $c --graphed 2>&1 | grep -v Warning | tail -6; done
