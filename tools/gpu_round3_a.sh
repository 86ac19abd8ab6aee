mkdir -p gpurun_out/g1
python -m pytest tests/test_gpu_ragged.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/g1/pytest_ragged.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g1/pytest_ragged.log
tail -15 gpurun_out/g1/pytest_ragged.log
python bench.py --config cfg4 --steps 20 --warmup 3 > gpurun_out/g1/cfg4_ragged.json 2> gpurun_out/g1/cfg4_ragged.err; tail -3 gpurun_out/g1/cfg4_ragged.err
python bench.py --config cfg4 --padded --steps 20 --warmup 3 > gpurun_out/g1/cfg4_padded.json 2> gpurun_out/g1/cfg4_padded.err; tail -3 gpurun_out/g1/cfg4_padded.err
python bench.py --config cfg5 --steps 20 --warmup 3 > gpurun_out/g1/cfg5.json 2> gpurun_out/g1/cfg5.err; tail -3 gpurun_out/g1/cfg5.err
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/g1/cfg2.json 2> gpurun_out/g1/cfg2.err; tail -3 gpurun_out/g1/cfg2.err
python bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline --graphs 4096 > gpurun_out/g1/cfg2_4096.json 2>&1
python - <<'P'
import json
for n in ("cfg4_ragged","cfg4_padded","cfg5","cfg2"):
    try:
        d=json.loads(open("gpurun_out/g1/%s.json"%n).read().strip().splitlines()[-1])
        print(n, "value %.4g ms/step %.4f"%(d["value"], d["ms_per_step"]), d["roofline"].get("kernel"), d["roofline"].get("frac"))
        for r in d["roofline"].get("per_call_table",[])[:40]:
            print("   %-28s %-44s x%d %8.1f us  hbm %.3f mfma %.3f"%(r["entry"],r["shape"],r["calls"],r["us"],r["frac_hbm"],r["frac_mfma_f32"]))
    except Exception as e: print(n, "ERR", e)
P
