"""Dev tool: the row-sharded mask threshold of tests/test_gpu_dist2.py, with the values printed."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparsebit_amd import ops, select, dist as sd
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    for trial in range(6):
        w = torch.randn(256, 96, generator=g).to(dev)
        idx = min(int(w.numel() * 0.5), w.numel() - 1)
        with sd.sharded_calibration():
            v = select.kth_values([w[rank::world].contiguous()], [[idx + 1]], ops.HipSelectBackend(), True, 0, False, dev)
        one = float(ops.kth_value(w, idx + 1, True))
        ref = float(torch.sort(w.abs().reshape(-1))[0][idx])
        print("rank", rank, "trial", trial, "sharded", float(v.reshape(())), "one-launch", one, "sort", ref, flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29777), nprocs=2, join=True)
