#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r06z; mkdir -p $OUT
for rep in 1 2; do
port=29711
for c in cfg2 cfg4 cfg5; do
  port=$((port+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --config $c --force-dist --no-cpu-baseline --steps 30 --warmup 5 2> $OUT/${c}_fd$rep.err | tail -1 > $OUT/${c}_fd$rep.json
  echo "$rep $c rc ${PIPESTATUS[0]} bytes $(wc -c < $OUT/${c}_fd$rep.json)"; grep -v 'Warning\|amdgpu.ids\|socket.cpp\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl' $OUT/${c}_fd$rep.err | tail -8
done
done
