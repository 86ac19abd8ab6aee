#!/bin/bash
# round 5, call w: the evidence at HEAD -- full GPU suite (accuracy record), rocprofv3 + PMC of every configuration, the traffic files
# put in place BEFORE the bench lines are taken (so that roofline.traffic belongs to these kernel sources), SURVEY 8(d) companions of
# cfg2, one-rank RCCL lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r05w; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
cp gpurun_out/accuracy_tests.json $OUT/accuracy_tests.json
bash tools/profile_round.sh r05w > /dev/null 2>&1
cp gpurun_out/prof_r05w/traffic_cfg2.json profiles/traffic_cfg2.json
for cfg in cfg4 cfg5 cfg3 cfg1; do
  bash tools/profile_config.sh r05w_$cfg 20 5 --config $cfg > /dev/null 2>&1
  cp gpurun_out/prof_r05w_$cfg/traffic.json profiles/traffic_$cfg.json
done
bash tools/bench_lines.sh r05w > $OUT/lines.txt 2>&1; cat $OUT/lines.txt
python bench.py --normalize --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/cfg2_normalize.json
python bench.py --graphs 4096 --graph --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/cfg2_graphs4096.json
port=29711
for c in cfg2 cfg4 cfg5; do
  port=$((port+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --config $c --force-dist --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 > $OUT/${c}_forcedist.json
done
python - <<'PY'
import json,os
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r05w"
for f in ("cfg2_normalize","cfg2_graphs4096","cfg2_forcedist","cfg4_forcedist","cfg5_forcedist"):
    try:
        d=json.loads(open(out+"/"+f+".json").read().strip().split("\n")[-1]); c=d.get("config",{}).get("collective") or d.get("collective")
        print(f, round(d["value"]), round(d["ms_per_step"],4), d.get("hipgraph_replay",{}).get("ms_per_step"), (json.dumps(c)[:300] if c else None))
    except Exception as e: print(f,"ERR",e)
PY
