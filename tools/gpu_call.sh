#!/bin/bash
# scratch script of the CURRENT gpurun call (rewritten per call; the named scripts -- evidence_round.sh, profile_round.sh,
# profile_config.sh, steps_all_configs.sh, bench_lines.sh -- are the ones that stay)
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; OUT=$REPO/gpurun_out/r06a; mkdir -p $OUT
{
  echo "== sensors"; ls /sys/class/drm/ 2>&1 | head; for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo $h; ls $h | tr '\n' ' '; echo; for f in freq1_input freq2_input power1_average power1_input; do [ -e $h/$f ] && echo "$f $(cat $h/$f)"; done; done
  python - <<'PY'
try:
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("amdsmi handles", len(hs))
    print(amdsmi.amdsmi_get_clock_info(hs[0], amdsmi.AmdSmiClkType.GFX))
    print(amdsmi.amdsmi_get_clock_info(hs[0], amdsmi.AmdSmiClkType.MEM))
    print(amdsmi.amdsmi_get_power_info(hs[0]))
except Exception as e:
    print("amdsmi failed", repr(e))
PY
} > $OUT/sensors.txt 2>&1
timeout 300 build/bwd_skeleton > $OUT/skeleton.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dense_bwd.py tests/test_gpu_parity.py -m gpu -q -x -k "one_pass_backward_against_fp64 or read_out_gradient or frozen_affine or graph_batch_normalization" > $OUT/pytest_new.log 2>&1; tail -3 $OUT/pytest_new.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
cat $OUT/skeleton.txt
