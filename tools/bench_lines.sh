#!/bin/bash
# the bench lines of every configuration (with the per-call roofline pass and the CPU baseline where the config has one) -> gpurun_out/lines_<tag>/
TAG=${1:-r04}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/lines_$TAG
python bench.py --steps 50 --warmup 5 2>/dev/null | tail -1 > gpurun_out/lines_$TAG/cfg2.json
for c in cfg4 cfg5 cfg3 cfg1; do python bench.py --config $c --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/lines_$TAG/$c.json; done
python - <<PY
import json
for c in ("cfg2","cfg4","cfg5","cfg3","cfg1"):
    d=json.load(open("gpurun_out/lines_$TAG/%s.json"%c)); r=d["roofline"]
    print(c, round(d["value"]), round(d["ms_per_step"],4), r.get("kernel"), r.get("bound"), r.get("frac"), r.get("traffic"), r.get("traffic_stale"), (d.get("cpu_baseline") or {}).get("value"))
PY
