"""Rehearsal of `bench.py --gpus 2` on a ONE-GPU box: both ranks on cuda:0, gloo instead of RCCL
(SBQ_BENCH_DEBUG_SINGLE_GPU=1).  The numbers mean nothing; what is asserted is the plumbing the driver's first real
SCALE run depends on: rendezvous, barrier + max-over-ranks timing, the statistic all-reduces, ONE JSON line from
rank 0 with the contract's keys."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_emit_one_valid_line():
    env = dict(os.environ, SBQ_BENCH_DEBUG_SINGLE_GPU="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--quick"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extras"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["unit"] == "elements/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert d["config"]["workload"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert d["parity_checked"] is True
    e = d["extras"]
    # the three statistic exchanges of the observers were timed at N = 2 (BASELINE: "observer all-reduce scaling")
    assert e["observer_allreduce_us"] > 0 and e["observer_allreduce_bytes"] == 4 * 4 * 4096
    assert e["observer_allreduce_mse_sum_us"] > 0 and e["observer_allreduce_mse_sum_bytes"] == 4096 * 80 * 8
    assert e["observer_allreduce_percentile_hist_sum_us"] > 0 and e["observer_allreduce_percentile_hist_sum_bytes"] == 2 * 2048 * 8
    # whole-job value: both ranks' elements over the max-over-ranks time
    assert abs(d["value"] - 2 * 5 * 4096 * 4096 / (d["ms_per_step"] * 5 * 1e-3)) <= 1e-6 * d["value"]


def test_bench_two_ranks_sharded_percentile_leg():
    """the N > 1 leg of config 3 (sharded percentile calibration, 12 observers in lock step): three collectives per
    MODEL (sample, one round for 16-bit activations, nothing else), bit-exact against the union"""
    env = dict(os.environ, SBQ_BENCH_DEBUG_SINGLE_GPU="1", SBQ_BENCH_SHARDED_LEG="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--quick"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    leg = d["extras"]["sharded_percentile_calibration"]
    assert leg["parity"] is True, leg
    assert leg["collectives_per_model"] <= 3 and leg["host_reads_per_model"] <= 1, leg
    assert leg["us_per_model"] > 0 and leg["bytes_per_model"] == 12 * (8193 + 4100) * 8


def test_bench_launches_itself_when_started_like_the_one_gpu_command():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the N = 1 command with another number) becomes
    its own torch.distributed.run launcher instead of dying on the WORLD_SIZE assertion (VERDICT r03 missing #2)."""
    env = dict(os.environ, SBQ_BENCH_DEBUG_SINGLE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--quick"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5
    rf = d["roofline"]
    # the roofline figure comes from windows of 1024 launches around the timed region, not from the 5 timed launches
    assert rf["windows_measured"] >= 12 and rf["kernel_avg_us_windows_min"] <= rf["kernel_avg_us"] <= rf["kernel_avg_us_windows_max"]
    assert rf["frac"] == rf["frac_1024_window_median"] and "frac_wall" in rf and "us_per_step_wall" in rf and "frac_rocprof" in rf
