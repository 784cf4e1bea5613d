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


class NumpyWindowBackend:
    """numpy stand-in for ops.HipWindowBackend: the four device steps of select.windowed_steps on fp32 keys.  The
    windows it derives differ from the HIP plan's (any bracket is valid: the protocol only trusts counts) -- what the
    gloo tests pin is the exchange protocol: every rank derives the SAME windows from the reduced sample, and the
    result equals the order statistic of the union."""

    PLAN_BINS, PLAN_SHIFT, BINS = 8192, 19, 2048

    def __init__(self, margin=4.0, stride=7):
        self.margin, self.stride = margin, stride  # a small margin makes windows MISS sometimes: extra rounds

    def expected_rounds(self):
        return 2

    def sample(self, shards, use_abs, device):
        self.use_abs = use_abs
        self.k = [NumpySelectBackend.keys(x.numpy().reshape(-1), use_abs) for x in shards]
        out = np.zeros(self.PLAN_BINS + 1, np.int64)
        for k in self.k:
            out[: self.PLAN_BINS] += np.bincount(k[:: self.stride] >> np.uint32(self.PLAN_SHIFT), minlength=self.PLAN_BINS)
            out[self.PLAN_BINS] += k.size
        return torch.from_numpy(out)

    def plan(self, sample, n_sel, percentile_alpha, ranks, device):
        h = sample.numpy()[: self.PLAN_BINS]
        n = int(sample[self.PLAN_BINS])
        S = int(h.sum())
        cum = np.cumsum(h)
        neg = int(h[: (0x80000000 >> self.PLAN_SHIFT)].sum())
        sels = []
        for s in range(n_sel):
            if percentile_alpha is not None:
                r = max(neg * percentile_alpha, 1.0 * S / max(n, 1)) if s == 0 else S - (S - neg) * percentile_alpha
                k = 0
            else:
                k = int(ranks[s])
                r = k * S / max(n, 1)
            m = self.margin * np.sqrt(max(r * (1 - min(r / max(S, 1), 1.0)), 1.0)) + 2
            a = int(np.searchsorted(cum, max(r - m, 1), side="left"))
            b = int(np.searchsorted(cum, min(r + m, S), side="left"))
            a, b = min(a, self.PLAN_BINS - 1), min(max(b, a), self.PLAN_BINS - 1)
            lo, width = a << self.PLAN_SHIFT, (b - a + 1) << self.PLAN_SHIFT
            sels.append(self._window(lo, width, k, fresh=True))
        return {"sel": sels, "n": n, "alpha": percentile_alpha, "vals": np.zeros(n_sel, np.float32)}

    def _window(self, lo, width, k, fresh):
        shift = 0
        while ((width + (1 << shift) - 1) >> shift) > self.BINS:
            shift += 1
        return {"lo": lo, "span": width - 1, "shift": shift, "k": k, "fresh": fresh, "done": False}

    def sweep(self, sel, shards, use_abs, count_signs):
        n_sel = len(sel["sel"])
        rec = np.zeros(2 * self.BINS + 4, np.int64)
        for k in self.k:
            k64 = k.astype(np.int64)
            for s, w in enumerate(sel["sel"]):
                if w["done"]:
                    continue
                d = k64 - w["lo"]
                inside = (d >= 0) & (d <= w["span"])
                rec[s * self.BINS:(s + 1) * self.BINS] += np.bincount(d[inside] >> w["shift"], minlength=self.BINS)
                rec[2 * self.BINS + s] += int((d < 0).sum())
            if count_signs:
                rec[2 * self.BINS + 2] += int((k < np.uint32(0x80000000)).sum())   # x < 0 (keys below key(+0))
                rec[2 * self.BINS + 3] += int((k == np.uint32(0xFFFFFFFF)).sum())  # NaN
        return torch.from_numpy(rec)

    def advance(self, sel, rec):
        rec = rec.numpy()
        n = sel["n"]
        for s, w in enumerate(sel["sel"]):
            if w["done"]:
                continue
            bins = rec[s * self.BINS:(s + 1) * self.BINS]
            below, total = int(rec[2 * self.BINS + s]), int(bins.sum())
            k = w["k"]
            if w["fresh"]:
                if sel["alpha"] is not None:
                    neg, nan = int(rec[2 * self.BINS + 2]), int(rec[2 * self.BINS + 3])
                    sel["neg"], sel["pos"] = neg, n - neg - nan
                    k = max(round(neg * sel["alpha"]), 1) if s == 0 else n - max(round(sel["pos"] * sel["alpha"]), 0)
                    k = min(max(k, 1), n)
                hi = w["lo"] + w["span"] + 1
                if k <= below:  # the window missed: everything below it
                    sel["sel"][s] = self._window(0, w["lo"], k, fresh=False)
                    continue
                if k > below + total:  # ... or everything above it
                    sel["sel"][s] = self._window(hi, (1 << 32) - hi, k - below - total, fresh=False)
                    continue
                k -= below
            cum = np.cumsum(bins)
            b = int(np.searchsorted(cum, k, side="left"))
            k -= int(cum[b - 1]) if b > 0 else 0
            lo = w["lo"] + (b << w["shift"])
            if w["shift"] == 0:
                u = np.array([lo], np.uint32)
                v = np.where((u & np.uint32(0x80000000)) != 0, u & np.uint32(0x7FFFFFFF), ~u).astype(np.uint32).view(np.float32)[0]
                if sel["alpha"] is not None:
                    v = v if (sel["neg"] if s == 0 else sel["pos"]) > 0 else np.float32(0)
                sel["vals"][s] = v
                w["done"] = True
            else:
                sel["sel"][s] = self._window(lo, min(1 << w["shift"], w["lo"] + w["span"] + 1 - lo), k, fresh=False)
        return torch.tensor([int(w["done"]) for w in sel["sel"]] + [1] * (2 - len(sel["sel"])), dtype=torch.int32)

    def values(self, sel):
        return torch.from_numpy(sel["vals"].copy())


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
    # ---- the windowed protocol (whole-tensor selections): sample SUM, plan, (sweep, SUM, advance) rounds ----
    with sd.sharded_calibration():
        flat_all = everything.reshape(-1)
        for alpha in (0.01, 0.2, 1e-4):
            sd.reset_stats()
            vals = sd.run_lockstep([select.windowed_steps([x.reshape(-1) for x in mine], NumpyWindowBackend(), torch.device("cpu"),
                                                          percentile_alpha=alpha)])[0]
            rmn_p, rmx_p = O.percentile(flat_all, alpha, 0, False)
            ok["win_pct_%g" % alpha] = float(vals[0]) == float(rmn_p[0]) and float(vals[1]) == float(rmx_p[0])
            ok["win_pct_collectives_%g" % alpha] = 3 <= sd.stats["collectives"] <= 6 and sd.stats["host_reads"] >= 1
        srt = np.sort(np.abs(flat_all))
        for kk in (1, flat_all.size // 2 + 1, flat_all.size):
            v = sd.run_lockstep([select.windowed_steps([x.reshape(-1) for x in mine], NumpyWindowBackend(margin=0.3), torch.device("cpu"),
                                                       use_abs=True, ranks=[kk])])[0]
            ok["win_kth_%d" % kk] = float(v[0]) == float(srt[kk - 1])  # (margin 0.3: windows miss, extra rounds, same answer)
        # a rank WITHOUT data takes part with zero records
        lone = [x.reshape(-1) for x in batches] if rank == 0 else []
        vals = sd.run_lockstep([select.windowed_steps(lone, NumpyWindowBackend(), torch.device("cpu"), percentile_alpha=0.01)])[0]
        rmn_p, rmx_p = O.percentile(flat_all, 0.01, 0, False)
        ok["win_empty_rank"] = float(vals[0]) == float(rmn_p[0]) and float(vals[1]) == float(rmx_p[0])
        # ---- a MODEL's worth of observers in lock step: the collectives are per model, not per quantizer ----
        def minmax_gen(scale):
            loc = [O.minmax(x.numpy() * scale, 1, False) for x in mine]
            mn, mx = yield ("max", (torch.from_numpy(np.min([l[0] for l in loc], 0)), torch.from_numpy(np.max([l[1] for l in loc], 0))))
            return mn, mx

        def mse_like_gen(scale):
            mn, mx = yield from minmax_gen(scale)
            buf = torch.zeros(81, dtype=torch.float64)
            buf[:80] = float(rank + 1) * scale
            buf[80] = float(sum(x.numel() for x in mine))
            buf = yield ("sum", buf)
            return mn, mx, buf

        sd.reset_stats()
        gens = [select.windowed_steps([x.reshape(-1) * sc for x in mine], NumpyWindowBackend(), torch.device("cpu"), percentile_alpha=0.05)
                for sc in (1.0, 2.0, 0.5)]
        gens += [minmax_gen(1.0), minmax_gen(3.0), mse_like_gen(2.0)]
        gens.append(select.kth_values_steps(mine, None, NumpySelectBackend(), False, 1, True, torch.device("cpu"),
                                            percentile_alpha=0.05, n_channels=6))
        res = sd.run_lockstep(gens)
        good = True
        for sc, vals in zip((1.0, 2.0, 0.5), res[:3]):
            rmn_p, rmx_p = O.percentile(flat_all * np.float32(sc), 0.05, 0, False)
            good &= float(vals[0]) == float(rmn_p[0]) and float(vals[1]) == float(rmx_p[0])
        for sc, (mn, mx) in zip((1.0, 3.0), res[3:5]):
            rmn, rmx = O.minmax(everything * sc, 1, False)
            good &= np.array_equal(mn.numpy(), rmn) and np.array_equal(mx.numpy(), rmx)
        mn, mx, buf = res[5]
        good &= float(buf[80]) == everything.size and float(buf[0]) == 2.0 * sum(r + 1 for r in range(world))
        rows = np.moveaxis(everything, 1, 0).reshape(6, -1)
        rmn_p, rmx_p = O.percentile(rows, 0.05, 0, True)
        vals2, _ = res[6]
        good &= np.array_equal(vals2[:, 0].numpy(), rmn_p) and np.array_equal(vals2[:, 1].numpy(), rmx_p)
        ok["lockstep_results"] = bool(good)
        # 7 observers: step 0 = {MAX (3 observers), int64 SUM (3 samples + 1 fixed-digit histogram)}, step 1 = {fp64 SUM,
        # int64 SUM}, step 2 = {int64 SUM}, then at most a few more int64 rounds for windows that missed; one kind of
        # exchange per step whatever the number of observers
        ok["lockstep_collectives"] = 5 <= sd.stats["collectives"] <= 9
        ok["lockstep_host_reads"] = 1 <= sd.stats["host_reads"] <= 4
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
        assert len(ok) >= 26


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
