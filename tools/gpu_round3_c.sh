mkdir -p gpurun_out/g3
python -m pytest tests -x -q -m gpu > gpurun_out/g3/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/g3/pytest_gpu.log
tail -8 gpurun_out/g3/pytest_gpu.log
for c in "cfg4" "cfg5"; do
  n=$(echo $c | tr -d ' -')
  python bench.py --config $c --steps 20 --warmup 3 > gpurun_out/g3/$n.json 2> gpurun_out/g3/$n.err; tail -2 gpurun_out/g3/$n.err
done
python - <<'P'
import json
for n in ("cfg4","cfg5"):
    try:
        d=json.loads(open("gpurun_out/g3/%s.json"%n).read().strip().splitlines()[-1])
        rows=d["roofline"].get("per_call_table",[])
        print(n, "value %.4g ms/step %.4f"%(d["value"], d["ms_per_step"]), d["roofline"].get("kernel"), d["roofline"].get("frac"), "abi us sum %.1f"%sum(r["us"] for r in rows))
        for r in rows[:25]:
            print("   %-28s %-44s x%d %8.1f us  hbm %.3f mfma %.3f"%(r["entry"],r["shape"],r["calls"],r["us"],r["frac_hbm"],r["frac_mfma_f32"]))
    except Exception as e: print(n, "ERR", e)
P
