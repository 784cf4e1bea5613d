"""Dev tool: two processes on one GPU, both running RESIDENT selections (ReLU data, extreme ranks) at the same time.
include/sbq.h says such launches must not run concurrently on one device; this shows what happens when they do."""
import os, sys, time
import torch, torch.multiprocessing as mp
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
def worker(rank, iters):
    sys.path.insert(0, ROOT)
    from sparsebit_amd import ops
    g = torch.Generator().manual_seed(rank)
    x = torch.relu(torch.randn(4096 * 4096, generator=g)).bfloat16().cuda()
    ref = torch.sort(x.float())[0]
    n = x.numel()
    t0 = time.time()
    bad = 0
    for i in range(iters):
        k = [1, n, n // 3, n // 2][i % 4]
        v = float(ops.kth_value(x, k, False))
        bad += v != float(ref[k - 1])
        mn, mx = ops.percentile_select([x.reshape(1, -1)], 1e-5, per_channel=False)
    torch.cuda.synchronize()
    print("rank", rank, "done", iters, "iterations in %.2f s, mismatches %d" % (time.time() - t0, bad), flush=True)
if __name__ == "__main__":
    mp.spawn(worker, args=(int(sys.argv[1]) if len(sys.argv) > 1 else 300,), nprocs=2, join=True)
