"""bench.py's multi-rank launcher on CPU: `python bench.py --gpus N` with no launcher environment must spawn its own ranks
(torch.distributed.run, rendezvous on 127.0.0.1) and rank 0 must print ONE JSON line.  The kernels need a GPU, so the
exchange runs in --dry mode (dummy gradients of the configuration's parameter shapes through kgcn_amd.parallel.GradBucket
over gloo); without --dry every spawned rank must stop at "needs a GPU" -- i.e. the launcher itself worked."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
              "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    return env


def _run(*args, timeout=240):
    return subprocess.run([sys.executable, BENCH] + list(args), env=_clean_env(), stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("config,scaling,graphs,ranks", [("cfg2", "weak", 1000, 2), ("cfg4", "weak", 0, 2), ("cfg2", "strong", 25, 2),
                                                         ("cfg1", "weak", 0, 2), ("cfg3", "weak", 0, 2), ("cfg5", "weak", 0, 2),
                                                         ("cfg2", "strong", 27, 8), ("cfg4", "weak", 0, 8)])
def test_bench_spawns_its_ranks_and_prints_one_line(config, scaling, graphs, ranks):
    args = ["--gpus", str(ranks), "--dry", "--device", "cpu", "--backend", "gloo", "--steps", "3", "--warmup", "1",
            "--config", config, "--scaling", scaling]
    if graphs:
        args += ["--graphs", str(graphs)]
    r = _run(*args)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == ranks and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == scaling
    assert res["config"]["dry"] is True and res["config"]["parallelism"] == "dp%d" % ranks
    assert res["config"]["exchanges_checked"] >= 3          # every exchange compared with the analytic weighted mean
    col = res["collective"]
    assert col["ranks"] == ranks and col["backend"] == "gloo" and col["bucket_floats"] > 0
    assert col["allreduce_us_standalone"]["median"] > 0
    assert len(col["per_rank_ms_per_step"]["all"]) == ranks
    eff = col["efficiency_expectation"]                     # what the exchange costs a step: bounds the N-rank loss
    assert 0 < eff["weak"]["at_8_ranks"] <= 1 and eff["exchange_us"] > 0 and 0 < eff["exchange_over_step"]
    assert (eff["strong"]["at_8_ranks"] is not None) == (config == "cfg2")
    assert set(col["env"]) == {"HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG"} and col["env"]["NCCL_DEBUG"] == "WARN"
    assert res["config"]["library"]["dev_overrides"] == {}


def test_bench_launch_command_is_the_documented_one():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5] == BENCH


def test_bench_without_a_gpu_fails_inside_the_spawned_ranks():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the real path runs instead")
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "rank 0 of 2" in r.stderr and "rank 1 of 2" in r.stderr and "needs a GPU" in r.stderr
    assert "the 2-rank launch failed" in r.stderr


def test_a_failing_rank_is_named_in_the_launchers_stderr():
    """a rank that dies with an exception (here: rank 1 cannot parse its poisoned environment) prints its traceback tagged with
    its rank id before the launcher tears the others down"""
    env = _clean_env()
    env["KGCN_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry", "--device", "cpu", "--backend", "gloo", "--steps", "1",
                        "--warmup", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240, cwd=ROOT)
    assert r.returncode != 0
    assert "[bench.py rank 1/2]" in r.stderr and "KGCN_BENCH_FAIL_RANK" in r.stderr


def test_bench_refuses_a_mismatched_launcher():
    env = _clean_env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry", "--device", "cpu", "--backend", "gloo"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
