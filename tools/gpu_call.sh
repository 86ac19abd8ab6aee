#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out/r06h
timeout 300 build/bwd_skeleton > gpurun_out/r06h/skeleton.txt 2>&1; grep 'H  pairs' gpurun_out/r06h/skeleton.txt
