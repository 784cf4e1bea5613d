"""Multi-rank path on CPU: world_size 2 over gloo.

sparsebit_amd.dist and sparsebit_amd.select only call torch.distributed, so these tests run
the very code the RCCL path runs on MI355X; per-rank statistics come from the CPU oracle
(tests may use it), standing in for the HIP kernels.  The claim under test is exactness:
sharded calibration == single-process calibration on the union of the shards.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpySelectBackend:
    """numpy stand-in for ops.HipSelectBackend (same protocol, include/sbq.h section 4)."""

    BINS = 2048

    @staticmethod
    def keys(x, use_abs):
        f = np.ascontiguousarray(x, dtype=np.float32)
        u = f.view(np.uint32).copy()
        if use_abs:
            u &= np.uint32(0x7FFFFFFF)
        u[u == np.uint32(0x80000000)] = 0
        neg = (u & np.uint32(0x80000000)) != 0
        k = np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)
        k[np.isnan(f)] = np.uint32(0xFFFFFFFF)
        return k

    def new_state(self, ranks, device):
        st = torch.zeros((len(ranks), len(ranks[0]), 2), dtype=torch.int64)
        st[:, :, 1] = torch.tensor(ranks, dtype=torch.int64)
        return st

    def new_hist(self, C, n_sel, device):
        return torch.zeros((C, n_sel, self.BINS), dtype=torch.int64)

    def zero_state(self, C, n_sel, device):
        return torch.zeros((C, n_sel, 2), dtype=torch.int64)

    def percentile_ranks(self, hist, state, alpha, C):
        """sbq_percentile_ranks: counts and ranks from the pass-0 histogram (percentile.py:27-43)"""
        h = hist[:, 0].numpy()
        neg, pos, nan = h[:, :1024].sum(1), h[:, 1024:2047].sum(1), h[:, 2047]
        n = neg + pos + nan
        for c in range(C):
            k_min = max(round(int(neg[c]) * alpha), 1)
            k_max = int(n[c]) - max(round(int(pos[c]) * alpha), 0)
            state[c, 0, 0], state[c, 0, 1] = 0, min(max(k_min, 1), int(n[c]))
            state[c, 1, 0], state[c, 1, 1] = 0, min(max(k_max, 1), int(n[c]))
        return torch.from_numpy(np.stack([neg, pos]).astype(np.int64))

    def histogram(self, x, state, hist, p, n_sel, use_abs, ch_axis, per_channel):
        x = x.numpy()
        rows = np.moveaxis(x, ch_axis, 0).reshape(x.shape[ch_axis], -1) if per_channel else x.reshape(1, -1)
        shift = (21, 10, 0)[p]
        known = (0, 0xFFE00000, 0xFFFFFC00)[p]
        dmask = 1023 if p == 2 else 2047
        for c in range(rows.shape[0]):
            k = self.keys(rows[c], use_abs)
            for s in range(n_sel):
                pre = int(state[c, s, 0])
                sel = k[(k & np.uint32(known)) == np.uint32(pre)]
                d = (sel >> np.uint32(shift)) & np.uint32(dmask)
                hist[c, s] += torch.from_numpy(np.bincount(d, minlength=self.BINS).astype(np.int64))

    def advance(self, hist, state, p, n_sel, C):
        shift = (21, 10, 0)[p]
        for c in range(C):
            for s in range(n_sel):
                k = int(state[c, s, 1])
                cum = np.cumsum(hist[c, s].numpy())
                b = int(np.searchsorted(cum, k, side="left"))
                state[c, s, 1] = k - (int(cum[b - 1]) if b > 0 else 0)
                state[c, s, 0] = int(state[c, s, 0]) | (b << shift)

    def finish(self, state, n_sel, C, use_abs):
        k = state[:, :, 0].numpy().astype(np.uint32)
        u = np.where((k & np.uint32(0x80000000)) != 0, k & np.uint32(0x7FFFFFFF), ~k).astype(np.uint32)
        return torch.from_numpy(u.view(np.float32).copy())


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from sparsebit_amd import dist as sd
    from sparsebit_amd import select

    g = torch.Generator().manual_seed(100)  # same stream on every rank: identical "global" data
    batches = [torch.randn(4, 6, 5, 5, generator=g).bfloat16().float() for _ in range(4)]
    mine = batches[rank::world]
    everything = np.concatenate([b.numpy() for b in batches], 0)
    ok = {}

    # inactive by default: nothing is exchanged
    a, b = sd.allreduce_minmax(torch.tensor([1.0]), torch.tensor([2.0]))
    ok["inactive"] = (not sd.active()) and a.item() == 1.0 and b.item() == 2.0

    with sd.sharded_calibration():
        assert sd.active() and sd.world_size() == world
        # ---- min/max, per tensor and per channel (NCHW, ch_axis 1) ----
        for perch in (False, True):
            loc = [O.minmax(x.numpy(), 1, perch) for x in mine]
            mn = torch.from_numpy(np.min([l[0] for l in loc], 0))
            mx = torch.from_numpy(np.max([l[1] for l in loc], 0))
            mn, mx = sd.allreduce_minmax(mn, mx)
            rmn, rmx = O.minmax(everything, 1, perch)
            ok["minmax%d" % perch] = np.array_equal(mn.numpy(), rmn) and np.array_equal(mx.numpy(), rmx)
        # many observers, one collective
        pairs = []
        for k, perch in enumerate((False, True, False)):
            loc = [O.minmax(x.numpy() * (k + 1), 1, perch) for x in mine]
            pairs.append((torch.from_numpy(np.min([l[0] for l in loc], 0)), torch.from_numpy(np.max([l[1] for l in loc], 0))))
        red = sd.allreduce_minmax_many(pairs)
        good = True
        for k, perch in enumerate((False, True, False)):
            rmn, rmx = O.minmax(everything * (k + 1), 1, perch)
            good &= np.array_equal(red[k][0].numpy(), rmn) and np.array_equal(red[k][1].numpy(), rmx)
            good &= red[k][0].shape == pairs[k][0].shape
        ok["minmax_many"] = bool(good)
        # NaN on one rank must reach every rank
        v = torch.tensor([float("nan") if rank == 1 else 0.5, 1.0])
        mn, mx = sd.allreduce_minmax(v.clone(), v.clone())
        ok["nan"] = bool(torch.isnan(mn[0]) and torch.isnan(mx[0]) and mn[1] == 1.0)
        # ---- MSE: SUM of per-shard squared-error tables == table of the union ----
        rmn, rmx = O.minmax(everything, 1, False)
        sse = np.zeros((1, 80))
        for x in mine:
            # per-shard table with the GLOBAL min/max: shrink candidates must be identical everywhere
            big = np.concatenate([x.numpy().reshape(-1), rmn, rmx])
            _, _, _, t_with = O.mse(big, -128, 127, True, per_channel=False)
            _, _, _, t_pad = O.mse(np.concatenate([rmn, rmx]), -128, 127, True, per_channel=False)
            sse += t_with - t_pad
        t = torch.from_numpy(sse)
        sd.allreduce_sum_(t)
        n = sd.allreduce_count(sum(x.numel() for x in mine))
        _, _, rbest, rsse = O.mse(everything, -128, 127, True, per_channel=False)
        ok["mse_count"] = n == everything.size
        ok["mse_table"] = np.allclose(t.numpy(), rsse, rtol=1e-9, atol=1e-9)
        ok["mse_best"] = int(np.argmin((t.numpy() / n).astype(np.float32))) == int(rbest[0])
        # ---- percentile: exact distributed radix select (two ranks per channel) ----
        for perch in (False, True):
            C = 6 if perch else 1
            neg = np.zeros(C, np.int64)
            pos = np.zeros(C, np.int64)
            for x in mine:
                r = np.moveaxis(x.numpy(), 1, 0).reshape(6, -1) if perch else x.numpy().reshape(1, -1)
                neg += (r < 0).sum(1)
                pos += (r >= 0).sum(1)
            cnt = torch.from_numpy(np.stack([neg, pos]))
            sd.allreduce_sum_(cnt)
            ntot = sd.allreduce_count(sum(x.numel() // C for x in mine))
            alpha = 0.01
            ranks = [[max(round(int(cnt[0, c]) * alpha), 1), ntot - max(round(int(cnt[1, c]) * alpha), 0)] for c in range(C)]
            vals = select.kth_values(mine, ranks, NumpySelectBackend(), False, 1, perch, torch.device("cpu"))
            rows = np.moveaxis(everything, 1, 0).reshape(6, -1) if perch else everything.reshape(1, -1)
            rmn_p, rmx_p = O.percentile(rows, alpha, 0, True)
            ok["pct%d" % perch] = np.array_equal(vals[:, 0].numpy(), rmn_p) and np.array_equal(vals[:, 1].numpy(), rmx_p)
            # the observer's own route: ranks derived from the all-reduced first histogram, no count pass
            vals2, cnt2 = select.kth_values(mine, None, NumpySelectBackend(), False, 1, perch, torch.device("cpu"),
                                            percentile_alpha=alpha, n_channels=C)
            ok["pct_dev_ranks%d" % perch] = (torch.equal(vals2, vals) and torch.equal(cnt2, cnt))
        # ---- mask threshold of a row-sharded weight: k-th |w| over the union ----
        w = torch.randn(64, 33, generator=g)
        shard = w[rank::world].contiguous()
        idx = min(int(w.numel() * 0.5), w.numel() - 1)
        v = select.kth_values([shard], [[idx + 1]], NumpySelectBackend(), True, 0, False, torch.device("cpu"))
        _, rt = O.l1_mask(w.numpy(), 0.5)
        ok["mask_thresh"] = float(v.reshape(())) == float(rt)
    ok["disabled_again"] = not sd.active()
    torch.save(ok, os.path.join(tmp, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_statistics_equal_single_process(tmp_path):
    from oracle import oracle as O

    O.build()
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        ok = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r))
        bad = [k for k, v in ok.items() if not v]
        assert not bad, (r, bad)
        assert len(ok) >= 12


def test_numpy_select_backend_is_the_protocol(oracle):
    """single process: the three-pass protocol == sort, for both key modes"""
    from sparsebit_amd import select

    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 3000, generator=g)
    x[:, ::9] = 0.0
    x[0, 5] = -0.0
    for use_abs in (False, True):
        srt = np.sort(np.abs(x.numpy()) if use_abs else x.numpy(), axis=1)
        ranks = [[1, 3000], [17, 1500], [2999, 3], [1000, 1001], [1, 1]]
        vals = select.kth_values([x], ranks, NumpySelectBackend(), use_abs, 0, True, torch.device("cpu"))
        for c in range(5):
            for s in range(2):
                assert vals[c, s].item() == srt[c, ranks[c][s] - 1]
