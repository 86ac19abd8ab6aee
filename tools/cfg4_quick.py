#!/usr/bin/env python3
"""cfg4 model step (tools/config_bench.py's cfg4), eager and hipGraph-replayed, one JSON line -- for A/B runs of library
variants in one gpurun call (VBENCH=cfg4_quick tools/variants.sh run a b a b)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.argv = [sys.argv[0], "--none"]
import importlib.util
spec = importlib.util.spec_from_file_location("config_bench_mod", os.path.join(ROOT, "tools", "config_bench.py"))
src = open(os.path.join(ROOT, "tools", "config_bench.py")).read()
src = src[:src.index("res = {}\nargs = ")]
mod = {"__file__": os.path.join(ROOT, "tools", "config_bench.py"), "__name__": "config_bench_defs"}
exec(compile(src, "config_bench_defs", "exec"), mod)
graphs, step = mod["cfg4"]()
ms = [round(mod["timed"](step), 4) for _ in range(3)]
print(json.dumps({"eager_ms": ms, "hipgraph_ms": round(mod["graphed"](step), 4)}))
