#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
KGCN_PROBE_LIB=build/variants/libkgcn_probe.so timeout 300 python tools/pairs_probe.py 100000 2 2>&1 | tail -21
KGCN_PROBE_LIB=build/variants/libkgcn_probehot.so timeout 300 python tools/pairs_probe.py 100000 2 2>&1 | tail -21
