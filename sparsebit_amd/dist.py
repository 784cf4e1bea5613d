"""Cross-GPU exchange of observer statistics (one process per GPU, RCCL over xGMI).

The reference has no observer collective: every rank calibrates on its own data
and DDP later broadcasts rank 0's qparams (examples/quantization_aware_training/
imagenet1k/basecase/main.py:240-255).  Here calibration batches shard across the
ranks and the statistics -- not the data -- are combined:
  min/max      one MAX all-reduce of [max, -min]           (exact: order independent)
  MSE          + one SUM all-reduce of the fp64 [C, 80] squared-error table (+ the element count, same buffer)
  percentile   per tensor: one SUM of the ranks' SAMPLE histograms (every rank then derives the same windows), one
               SUM of the window counts per sweep -- one sweep for 16-bit data, two for fp32 (select.windowed_steps);
               per channel: one SUM of int64 histograms per fixed-digit pass (select.kth_values_steps); exact
  LSQ init     one SUM all-reduce of [sum|x|, count]
Messages are <= 2.6 MB, i.e. latency bound on xGMI; everything the observers of a MODEL need at one step of their
protocols travels in ONE flat buffer per kind (run_lockstep): per model, not per quantizer.

These helpers are device agnostic (they only call torch.distributed), so the
world_size-2 gloo tests exercise exactly the code the RCCL path runs.
"""
import contextlib

import torch
import torch.distributed as dist

_sync_enabled = False
_group = None
_even_alone = False


def enable(group=None):
    """Turn on statistic all-reduces inside observers (no-op when world_size == 1)."""
    global _sync_enabled, _group
    _sync_enabled = True
    _group = group


def disable():
    global _sync_enabled, _group
    _sync_enabled = False
    _group = None


@contextlib.contextmanager
def sharded_calibration(group=None):
    enable(group)
    try:
        yield
    finally:
        disable()


def collectives_even_alone(flag=True):
    """Issue the collectives on a ONE-rank communicator too (default: a lone rank skips them).  Every wire format
    then crosses RCCL on a single leased GPU -- tests/test_gpu_rccl_ws1.py, and the N = 1 point of bench.py's
    observer all-reduce latencies.  Results are unchanged by construction: a reduction over one rank is the identity."""
    global _even_alone
    _even_alone = bool(flag)


def active():
    if not (_sync_enabled and dist.is_available() and dist.is_initialized()):
        return False
    return _even_alone or dist.get_world_size(_group) > 1


def world_size():
    return dist.get_world_size(_group) if active() else 1


def init_single_rank_rccl(device):
    """A one-rank RCCL communicator on `device` (file-store rendezvous: no port, no environment) -- what a one-GPU
    lease can still put under the collective path.  Returns True when THIS call created the default group."""
    import atexit
    import os
    import tempfile

    if dist.is_initialized():
        return False
    # (HSA_ENABLE_IPC_MODE_LEGACY=0 -- dmabuf IPC, the only kind this pool's host driver supports -- has to be in the
    # environment BEFORE the HIP runtime initialises: the launcher's / the test's job, not this function's.  One rank
    # shares no memory across processes, so nothing here depends on it.)
    fd, path = tempfile.mkstemp(prefix="sbq_rccl_ws1_")
    os.close(fd)
    os.unlink(path)  # (FileStore creates it again)

    def _cleanup(p=path):
        try:
            os.unlink(p)
        except OSError:
            pass

    atexit.register(_cleanup)
    dist.init_process_group("nccl", init_method="file://" + path, world_size=1, rank=0, device_id=device)
    return True


def allreduce_minmax(min_val, max_val):
    """(min, max) over all ranks with ONE collective: MAX over [max, -min, isnan(max), isnan(min)].

    The NaN flags travel with the values because torch.min/max propagate NaN locally
    (and the reference's observers therefore do), while a MAX collective's NaN
    behaviour is backend defined."""
    if not active():
        return min_val, max_val
    if min_val.is_cuda:
        # device statistics: one pack kernel, one collective, one unpack kernel
        from . import ops

        buf = ops.minmax_pack(min_val, max_val)
        _all_reduce(buf, dist.ReduceOp.MAX)
        mn, mx = ops.minmax_unpack(buf, min_val.shape)
        return mn.to(min_val.dtype), mx.to(max_val.dtype)
    # host tensors (the gloo tests): the same wire format with torch ops
    n = min_val.numel()
    mx = max_val.reshape(-1).float()
    mn = min_val.reshape(-1).float()
    buf = torch.cat([torch.nan_to_num(mx, nan=float("-inf")), torch.nan_to_num(-mn, nan=float("-inf")),
                     torch.isnan(mx).float(), torch.isnan(mn).float()])
    _all_reduce(buf, dist.ReduceOp.MAX)
    nan = torch.full((n,), float("nan"), dtype=buf.dtype, device=buf.device)
    mx = torch.where(buf[2 * n:3 * n] > 0, nan, buf[:n])
    mn = torch.where(buf[3 * n:] > 0, nan, -buf[n:2 * n])
    return mn.reshape(min_val.shape).to(min_val.dtype), mx.reshape(max_val.shape).to(max_val.dtype)


def allreduce_minmax_many(pairs):
    """[(min, max), ...] of many observers combined with ONE collective (one flat buffer):
    the messages are latency bound, so a model's worth of min-max observers costs one
    all-reduce instead of one per quantizer.  Returns the reduced pairs, shapes preserved."""
    if not active() or not pairs:
        return list(pairs)
    sizes = [p[0].numel() for p in pairs]
    mn = torch.cat([p[0].reshape(-1).float() for p in pairs])
    mx = torch.cat([p[1].reshape(-1).float() for p in pairs])
    mn, mx = allreduce_minmax(mn, mx)
    out, o = [], 0
    for (a, b), n in zip(pairs, sizes):
        out.append((mn[o:o + n].reshape(a.shape).to(a.dtype), mx[o:o + n].reshape(b.shape).to(b.dtype)))
        o += n
    return out


def allreduce_sum_(t):
    """In-place SUM over ranks (fp64 / int64 tables: exact or order-insensitive enough)."""
    if active():
        _all_reduce(t, dist.ReduceOp.SUM)
    return t


def allreduce_count(n):
    """Python number summed over ranks.  (A host round trip: the observers carry their counts inside the
    buffers they all-reduce anyway -- this stays for callers outside the calibration protocol.)"""
    if not active():
        return n
    t = torch.tensor([float(n)], dtype=torch.float64, device=_comm_device())
    _all_reduce(t, dist.ReduceOp.SUM)
    stats["host_reads"] += 1
    return type(n)(t.item())


# what crossed the wire / the PCIe bus since the last reset_stats(): the tests count collectives per MODEL
stats = {"collectives": 0, "bytes": 0, "host_reads": 0}


def reset_stats():
    for k in stats:
        stats[k] = 0


def _all_reduce(t, op):
    dist.all_reduce(t, op=op, group=_group)
    stats["collectives"] += 1
    stats["bytes"] += t.numel() * t.element_size()


# ---- observers in lock step ---------------------------------------------------------------------------------
# A sharded observer is a GENERATOR: it computes its rank-local statistics, yields a request, and is resumed with the
# globally reduced answer --
#     ("max", (min, max))   ->  (min, max) over all ranks          [allreduce_minmax's wire format]
#     ("sum", tensor)       ->  the same tensor, summed in place   [int64 / float64]
#     ("host", tensor)      ->  its values as a Python list        [the one place a protocol may look at device data]
# and returns its result.  run_lockstep() advances MANY such generators together and sends the requests of one step
# that share a kind (and dtype) as ONE flat collective / ONE device-to-host copy: a model's worth of min-max, MSE and
# percentile observers costs the collectives of one observer of each kind (the messages are latency bound), instead
# of tools/calibration.py:102-115's per-quantizer loop turning into a per-quantizer exchange.  Every rank must drive
# the same generators in the same order -- they are built from the model, which is replicated -- and a generator's
# control flow may depend on reduced values only.
def run_lockstep(gens):
    """-> [return value of each generator].  Works without an initialised process group too (the requests are then
    answered locally), so the same code path serves one process."""
    n = len(gens)
    results = [None] * n
    reqs = {}

    def step(i, value):
        try:
            reqs[i] = gens[i].send(value)
        except StopIteration as e:
            results[i] = e.value

    for i in range(n):
        step(i, None)
    while reqs:
        cur, answers = reqs, {}
        reqs = {}
        groups = {}
        for i in sorted(cur):
            kind, payload = cur[i]
            dt = payload[0].dtype if kind == "max" else payload.dtype
            groups.setdefault((kind, str(dt)), []).append(i)
        for key in sorted(groups):  # the same order on every rank
            kind, members = key[0], groups[key]
            if kind == "max":
                red = allreduce_minmax_many([cur[i][1] for i in members])
                for i, pair in zip(members, red):
                    answers[i] = pair
            elif kind == "sum":
                ts = [cur[i][1] for i in members]
                if active():
                    if len(ts) == 1:
                        _all_reduce(ts[0], dist.ReduceOp.SUM)
                    else:
                        # one gather launch, one collective, one scatter launch (a copy_ per member was a launch per
                        # observer: 12 of them behind a ~20 us collective on DeiT-small)
                        flat = torch.cat([t.reshape(-1) for t in ts])
                        _all_reduce(flat, dist.ReduceOp.SUM)
                        parts = list(flat.split([t.numel() for t in ts]))
                        if all(t.is_contiguous() for t in ts):
                            torch._foreach_copy_([t.view(-1) for t in ts], parts)
                        else:
                            for t, part in zip(ts, parts):
                                t.copy_(part.reshape(t.shape))
                for i, t in zip(members, ts):
                    answers[i] = t
            elif kind == "host":
                ts = [cur[i][1].reshape(-1) for i in members]
                flat = (torch.cat(ts) if len(ts) > 1 else ts[0]).cpu()  # ONE device-to-host copy for all of them
                stats["host_reads"] += 1
                vals, o = flat.tolist(), 0
                for i, t in zip(members, ts):
                    answers[i] = vals[o:o + t.numel()]
                    o += t.numel()
            else:
                raise ValueError("unknown exchange %r" % (kind,))
        for i in sorted(answers):
            step(i, answers[i])
    return results


def _comm_device():
    if dist.get_backend(_group) == "nccl":  # RCCL on ROCm
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
