"""One-rank RCCL communicator on the leased GPU: every wire format of sparsebit_amd.dist through the real library.

  python tools/rccl_ws1.py            -> one JSON line: identity checks + per-collective latency (us)

A one-GPU lease cannot form a ring, but it CAN create a world_size-1 RCCL communicator: init_process_group("nccl"),
the device-side pack / unpack kernels around the MAX all-reduce, the fp64 and int64 SUM buffers and
HSA_ENABLE_IPC_MODE_LEGACY=0 then meet the library they were written for (the gloo tests never load it), and the
latencies are the N = 1 point of the "observer all-reduce scaling at 1/2/4/8 GPUs" curve of BASELINE.json -- the
launch + RCCL-kernel floor that the xGMI hops of N > 1 add to.  bench.py runs this file in a subprocess (with a
timeout: a hung rendezvous must not take the benchmark with it) and copies the numbers into extras.

The reference has no observer collective (examples/quantization_aware_training/imagenet1k/basecase/main.py:240-255
calibrates per rank); the formats are SURVEY.md 8(e)'s.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    from sparsebit_amd import dist as sd
    from sparsebit_amd import lib as L
    from sparsebit_amd import ops

    assert torch.cuda.is_available(), "needs the MI355X"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    t0 = time.perf_counter()
    sd.init_single_rank_rccl(dev)
    sd.collectives_even_alone(True)
    C = 4096
    g = torch.Generator().manual_seed(3)
    mn = (-torch.rand(C, generator=g) * 3).to(dev)
    mx = (torch.rand(C, generator=g) * 5).to(dev)
    # the cases the flag lanes exist for: NaN on either side, +-inf
    mn[7], mx[9], mn[11], mx[11] = float("nan"), float("nan"), float("-inf"), float("inf")
    sse = torch.rand(C, L.MSE_CANDIDATES, generator=g, dtype=torch.float64).to(dev)
    sse_cnt = torch.cat([sse.reshape(-1), torch.tensor([16777216.0], dtype=torch.float64, device=dev)])  # [C * 80 + 1]
    sample = torch.randint(0, 1 << 40, (L.DIST_SAMPLE_WORDS,), generator=g, dtype=torch.int64).to(dev)
    rnd = torch.randint(0, 1 << 40, (L.DIST_ROUND_WORDS,), generator=g, dtype=torch.int64).to(dev)
    hist = torch.randint(0, 1 << 30, (1, 2, L.RADIX_BINS), generator=g, dtype=torch.int64).to(dev)
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
           "hsa_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}

    def same(a, b):
        return bool(torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))
                    and torch.equal(torch.isnan(a), torch.isnan(b)))

    with sd.sharded_calibration():
        assert sd.active() and sd.world_size() == 1
        sd.reset_stats()
        mn2, mx2 = sd.allreduce_minmax(mn, mx)
        out["minmax_identity"] = same(mn2, mn) and same(mx2, mx)
        out["minmax_collectives"] = sd.stats["collectives"]
        out["minmax_bytes"] = sd.stats["bytes"]
        checks = {}
        for name, t in (("mse_sum", sse_cnt), ("sample_sum", sample), ("round_sum", rnd), ("hist_sum", hist)):
            ref = t.clone()
            sd.allreduce_sum_(t)
            checks[name] = bool(torch.equal(t, ref))
        out["sum_identity"] = checks
        torch.cuda.synchronize(dev)
        out["init_and_first_collectives_s"] = round(time.perf_counter() - t0, 2)

        stream = torch.cuda.current_stream(dev)

        def timed(fn, iters=200):
            for _ in range(20):
                fn()
            torch.cuda.synchronize(dev)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(iters):
                fn()
            b.record(stream)
            torch.cuda.synchronize(dev)
            return round(a.elapsed_time(b) * 1e3 / iters, 2)

        packed = ops.minmax_pack(mn, mx)
        lat = {
            # what an observer pays: pack kernel + MAX all-reduce + unpack kernel
            "minmax_pack_allreduce_unpack_us": timed(lambda: sd.allreduce_minmax(mn, mx)),
            "minmax_allreduce_only_us": timed(lambda: sd._all_reduce(packed, dist.ReduceOp.MAX)),
            "minmax_bytes": packed.numel() * packed.element_size(),
            "mse_sum_us": timed(lambda: sd.allreduce_sum_(sse_cnt)),
            "mse_sum_bytes": sse_cnt.numel() * 8,
            "percentile_sample_sum_us": timed(lambda: sd.allreduce_sum_(sample)),
            "percentile_sample_sum_bytes": sample.numel() * 8,
            "percentile_round_sum_us": timed(lambda: sd.allreduce_sum_(rnd)),
            "percentile_round_sum_bytes": rnd.numel() * 8,
            "percentile_hist_sum_us": timed(lambda: sd.allreduce_sum_(hist)),
            "percentile_hist_sum_bytes": hist.numel() * 8,
        }
        # run_lockstep's flat SUM of a model's worth of requests (12 percentile round records): gather + collective +
        # scatter, against the collective alone on a buffer of the same size
        recs = [rnd.clone() for _ in range(12)]
        flat12 = torch.cat(recs)

        def lockstep_sum():
            def gen(t):
                yield ("sum", t)

            sd.run_lockstep([gen(t) for t in recs])

        lat["lockstep_12_records_sum_us"] = timed(lockstep_sum, 100)
        lat["flat_12_records_allreduce_only_us"] = timed(lambda: sd._all_reduce(flat12, dist.ReduceOp.SUM), 100)
        out["latency"] = lat
    out["ok"] = bool(out["minmax_identity"] and all(checks.values()) and out["minmax_collectives"] == 1
                     and out["minmax_bytes"] == 16 * C)
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()
    return 0 if out["ok"] else 1


if __name__ == "__main__":
    sys.exit(main())
