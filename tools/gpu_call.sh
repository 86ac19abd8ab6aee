#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
time python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warning | tail -12
