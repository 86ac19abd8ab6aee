#!/bin/bash
# The judged evidence of a round at HEAD, one gpurun call:   bash tools/evidence_round.sh <tag> [notests]
#   1. the full GPU suite (accuracy record -> gpurun_out/<tag>/accuracy_tests.json + accuracy_fingerprint.json)
#   2. rocprofv3 kernel trace + PMC passes of every configuration (tools/profile_round.sh: cfg2, tools/profile_config.sh: the rest),
#      the traffic files put in place BEFORE the bench lines are taken (roofline.traffic then belongs to these kernel sources)
#   3. the bench line of every configuration (tools/bench_lines.sh)
#   4. SURVEY 8(d) companions of cfg2 (--normalize, --graphs 4096 --graph) and the one-rank RCCL lines of cfg2 / cfg4 / cfg5
# Copy what is to be judged from gpurun_out/ into profiles/<tag>_* afterwards (tools/collect_evidence.py <tag>).
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
if [ "${2:-}" != "notests" ]; then
  timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
  cp gpurun_out/accuracy_tests.json gpurun_out/accuracy_fingerprint.json $OUT/ 2>/dev/null
fi
bash tools/profile_round.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/traffic_cfg2.json profiles/traffic_cfg2.json
for cfg in cfg4 cfg5 cfg3 cfg1; do
  bash tools/profile_config.sh ${TAG}_$cfg 20 5 --config $cfg > /dev/null 2>&1
  cp gpurun_out/prof_${TAG}_$cfg/traffic.json profiles/traffic_$cfg.json
done
bash tools/bench_lines.sh $TAG > $OUT/lines.txt 2>&1; cat $OUT/lines.txt
python bench.py --normalize --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/cfg2_normalize.json
python bench.py --graphs 4096 --graph --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/cfg2_graphs4096.json
port=29711
for c in cfg2 cfg4 cfg5; do
  port=$((port+1))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 1 --config $c --force-dist --no-cpu-baseline --steps 30 --warmup 5 2> $OUT/${c}_forcedist.err | tail -1 > $OUT/${c}_forcedist.json
done
python - <<PY
import json
out="$OUT"
for f in ("cfg2_normalize","cfg2_graphs4096","cfg2_forcedist","cfg4_forcedist","cfg5_forcedist"):
    try:
        d=json.loads(open(out+"/"+f+".json").read().strip().split("\n")[-1]); c=d.get("config",{}).get("collective") or d.get("collective")
        print(f, round(d["value"]), round(d["ms_per_step"],4), d.get("hipgraph_replay",{}).get("ms_per_step"), (json.dumps(c)[:300] if c else None))
    except Exception as e: print(f,"ERR",e)
PY
# ---- round 6 additions: the backward's two forms on THIS box, the memory skeleton, the per-role probe, multi-channel Bconv, sweeps ----
if [ -d build/variants ]; then
  { for rep in 1 2; do VB_TIMING_ONLY=1 bash tools/variants.sh run planes pairs pairs_hot planes_hot 2>&1 | cut -c1-900; done; } > $OUT/backward_forms.txt 2>&1
  { KGCN_PROBE_LIB=build/variants/libkgcn_probe.so timeout 300 python tools/pairs_probe.py 100000 2; KGCN_PROBE_LIB=build/variants/libkgcn_probehot.so timeout 300 python tools/pairs_probe.py 100000 2; } > $OUT/pairs_probe.txt 2>&1
fi
[ -x build/bwd_skeleton ] && timeout 300 build/bwd_skeleton > $OUT/skeleton.txt 2>&1
timeout 600 python tools/bconv_bench.py > $OUT/bconv_c6.jsonl 2> $OUT/bconv.err
timeout 900 python tools/fuzz_gpu.py 300 606 > $OUT/fuzz_gpu.txt 2>&1; tail -3 $OUT/fuzz_gpu.txt
timeout 900 python tools/fuzz_gemmh.py 200 606 > $OUT/fuzz_gemmh.txt 2>&1; tail -2 $OUT/fuzz_gemmh.txt
