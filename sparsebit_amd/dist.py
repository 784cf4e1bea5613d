"""Cross-GPU exchange of observer statistics (one process per GPU, RCCL over xGMI).

The reference has no observer collective: every rank calibrates on its own data
and DDP later broadcasts rank 0's qparams (examples/quantization_aware_training/
imagenet1k/basecase/main.py:240-255).  Here calibration batches shard across the
ranks and the statistics -- not the data -- are combined:
  min/max      one MAX all-reduce of [max, -min]           (exact: order independent)
  MSE          + one SUM all-reduce of the fp64 [C, 80] squared-error table
  percentile   one SUM all-reduce of int64 histograms per radix pass (exact)
  LSQ init     one SUM all-reduce of [sum|x|, count]
Messages are <= 1.3 MB, i.e. latency bound on xGMI; everything a quantizer needs
travels in ONE flat buffer per step.

These helpers are device agnostic (they only call torch.distributed), so the
world_size-2 gloo tests exercise exactly the code the RCCL path runs.
"""
import contextlib

import torch
import torch.distributed as dist

_sync_enabled = False
_group = None


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


def active():
    return _sync_enabled and dist.is_available() and dist.is_initialized() and dist.get_world_size(_group) > 1


def world_size():
    return dist.get_world_size(_group) if active() else 1


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
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=_group)
        mn, mx = ops.minmax_unpack(buf, min_val.shape)
        return mn.to(min_val.dtype), mx.to(max_val.dtype)
    # host tensors (the gloo tests): the same wire format with torch ops
    n = min_val.numel()
    mx = max_val.reshape(-1).float()
    mn = min_val.reshape(-1).float()
    buf = torch.cat([torch.nan_to_num(mx, nan=float("-inf")), torch.nan_to_num(-mn, nan=float("-inf")),
                     torch.isnan(mx).float(), torch.isnan(mn).float()])
    dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=_group)
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
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group)
    return t


def allreduce_count(n):
    """Python number summed over ranks."""
    if not active():
        return n
    t = torch.tensor([float(n)], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=_group)
    return type(n)(t.item())


def _comm_device():
    if dist.get_backend(_group) == "nccl":  # RCCL on ROCm
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
