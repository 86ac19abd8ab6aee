#!/bin/bash
# scratch script of the CURRENT gpurun call (rewritten per call; the named scripts -- evidence_round.sh, profile_round.sh,
# profile_config.sh, steps_all_configs.sh, bench_lines.sh -- are the ones that stay).  Default: the GPU suite, smoke() and the bench line.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/final; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cp gpurun_out/accuracy_tests.json gpurun_out/accuracy_fingerprint.json $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warning | tail -3
python bench.py > $OUT/bench.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/final/bench.json').read().strip().split('\n')[-1]); r=d['roofline']
print(d['metric'], d['value'], d['ms_per_step'], 'frac', r['frac'], 'probe', r.get('frac_of_probe_rate',{}).get('value'), 'sclk', r['box']['sustained']['sclk_mhz'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
