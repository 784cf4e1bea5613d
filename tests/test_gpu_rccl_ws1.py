"""RCCL under the collective path, on ONE leased GPU.

Every multi-rank test of this repo runs over gloo (tests/test_dist_gloo.py on CPU statistics, tests/test_gpu_dist2.py
with two processes on cuda:0): the library the exchange is written for -- RCCL, torch.distributed's "nccl" backend on
ROCm -- never saw it.  A one-GPU lease cannot form a ring, but it can create a world_size-1 RCCL communicator:

  * test_product_call_sites_over_rccl_world_size_1: tests/test_gpu_dist2.py's whole worker -- every observer class,
    DeviceCalibrator.calibrate(sharded=True), the sparser's sharded threshold, the windowed and the fixed-digit
    selection protocols, the lock-step model calibration with its collective counts -- with init_process_group("nccl",
    world_size=1, device_id=cuda:0) and dist.collectives_even_alone(): every MAX / fp64 SUM / int64 SUM is issued on
    device tensors and must leave the single-process result bit for bit (a reduction over one rank is the identity);
  * test_wire_formats_and_latency: tools/rccl_ws1.py -- the packed [4C] fp32 MAX buffer with NaN flags and +-inf, the
    fp64 [C*80+1] SUM, the int64 SBQ_DIST_SAMPLE_WORDS / SBQ_DIST_ROUND_WORDS SUMs -- plus the per-collective latency
    that bench.py reports as the N = 1 point of the observer all-reduce curve.

Both run in subprocesses with a timeout: a hung rendezvous fails the test instead of the session.
Contract: SURVEY.md 8(e); the reference has no observer collective (examples/quantization_aware_training/imagenet1k/
basecase/main.py:240-255).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = os.pathsep.join([ROOT, HERE, env.get("PYTHONPATH", "")])
    return env


def test_product_call_sites_over_rccl_world_size_1(tmp_path):
    code = ("import sys, torch; sys.path[:0] = [%r, %r]; import test_gpu_dist2 as T; "
            "T._worker(0, 1, 0, %r, backend='nccl')" % (ROOT, HERE, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ok = torch.load(os.path.join(str(tmp_path), "rank0.pt"))
    bad = [k for k, v in ok.items() if not v]
    assert not bad, bad
    from test_gpu_dist2 import CASES

    assert len(ok) >= len(CASES) + 3 + 18 + 5 + 6


def test_wire_formats_and_latency():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_ws1.py")], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["ok"] and rec["backend"] == "nccl" and rec["world_size"] == 1
    assert rec["minmax_identity"] and all(rec["sum_identity"].values())
    lat = rec["latency"]
    for k in ("minmax_pack_allreduce_unpack_us", "mse_sum_us", "percentile_sample_sum_us", "percentile_round_sum_us",
              "percentile_hist_sum_us", "lockstep_12_records_sum_us"):
        assert lat[k] > 0.0, (k, lat)
