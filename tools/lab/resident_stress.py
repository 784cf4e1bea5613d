"""Round 5: tests/test_gpu_r03.py::test_concurrent_resident_selections_neither_hang_nor_differ failed ONCE in a full
-m gpu run (1 of 3 full runs, 0 of 8 isolated ones).  The same two-process stress, many rounds, with every mismatch
printed (which call, which rank, got / want).   python tools/lab/resident_stress.py [rounds] [iters]"""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")


def worker(rank, rounds, iters):
    sys.path.insert(0, ROOT)
    if os.environ.get("SBQ_LIB"):  # a variant library (tools/lab/build_variant.py -DSBQ_RESIGN_TICKS=50ull: resign at once)
        from sparsebit_amd import lib as L

        L.LIB_PATH = os.environ["SBQ_LIB"]
    from sparsebit_amd import ops

    g = torch.Generator().manual_seed(100 + rank)
    x = torch.relu(torch.randn(4096 * 4096, generator=g)).bfloat16().cuda()
    ref = torch.sort(x.float())[0]
    n = x.numel()
    bad = 0
    for r in range(rounds):
        for i in range(iters):
            k = [1, n, n // 3, n // 2][i % 4]
            v = float(ops.kth_value(x, k, False))
            if v != float(ref[k - 1]):
                bad += 1
                print("rank %d round %d iter %d: kth(%d) = %r, want %r" % (rank, r, i, k, v, float(ref[k - 1])), flush=True)
            mn, mx = ops.percentile_select([x.reshape(1, -1)], 1e-5, per_channel=False)
            want = float(ref[n - max(round(n * 1e-5), 0) - 1])
            if float(mn) != 0.0 or float(mx) != want:
                bad += 1
                print("rank %d round %d iter %d: percentile = (%r, %r), want (0.0, %r)" % (rank, r, i, float(mn), float(mx), want), flush=True)
    torch.cuda.synchronize()
    print("rank %d: %d mismatches in %d x %d iterations" % (rank, bad, rounds, iters), flush=True)


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    mp.spawn(worker, args=(rounds, iters), nprocs=2, join=True)
