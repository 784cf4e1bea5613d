"""Dev tool: state of the one-launch selection engine after its first (grid-wide) round -- knob 1 == 777 skips the
lonely rounds and the clean-up, so the selector state stays in the workspace."""
import os, sys, struct
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
g = torch.Generator().manual_seed(0)
R = C = 4096
w = (torch.randn(R, C, generator=g) * torch.logspace(-2, 1, R).unsqueeze(1)).bfloat16().to(dev)
n = R * C
OLD = 256 + 64 * 128 + 8 * 2 * 2048 * 4 + 256
for k in (n // 2 + 1, 1, n):
    sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
    out = torch.full((2,), -1.0, dtype=torch.float32, device=dev)
    L.set_tuning(1, 777)
    rc = lib.sbq_kth_value(L.ptr(w), L.BF16, n, 1, k, L.ptr(out), L.ptr(sw), sw.numel(), st)
    torch.cuda.synchronize()
    L.set_tuning(1, 0)
    raw = bytes(sw[OLD:OLD + 96].cpu().numpy())
    for s in range(2):
        lo, shift, span, side, kk, done, fresh = struct.unpack_from("<IIIIqII", raw, 32 * s)
        print("k=%d sel%d: lo=%08x shift=%d span=%08x side=%d k=%d done=%d fresh=%d" % (k, s, lo, shift, span, side, kk, done, fresh))
    nn, arrivals = struct.unpack_from("<qI", raw, 64)
    slots = sw[OLD + 256: OLD + 256 + 64 * 128].view(torch.int64).reshape(64, 16).cpu()
    hist = sw[OLD + 256 + 64 * 128: OLD + 256 + 64 * 128 + 8 * 2 * 2048 * 4].view(torch.int32).reshape(8, 2, 2048).cpu()
    ref = torch.sort(w.float().abs().reshape(-1))[0][k - 1].item()
    print("  n=%d arrivals=%d out=%r ref=%r  slots.below sum=%d  hist sum=%d nonzero bins=%d" % (
        nn, arrivals, out.tolist(), ref, int(slots[:, 0].sum()), int(hist.sum()), int((hist != 0).sum())))

print("---- normal mode, one call at a time")
import time
sw = torch.zeros(lib.sbq_radix_select_workspace_bytes(1, 2), dtype=torch.uint8, device=dev)
out = torch.full((2,), -1.0, dtype=torch.float32, device=dev)
for it in range(6):
    k = n // 2 + 1
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    rc = lib.sbq_kth_value(L.ptr(w), L.BF16, n, 1, k, L.ptr(out), L.ptr(sw), sw.numel(), st)
    b.record(); torch.cuda.synchronize()
    nz = int((sw[OLD:] != 0).sum())
    print("call %d: rc=%d %.1f us out=%r nonzero workspace bytes after=%d" % (it, rc, a.elapsed_time(b) * 1e3, out[0].item(), nz))

print("---- back to back")
xs = [w] + [torch.roll(w, i, 1).contiguous() for i in range(1, 4)]
def loop(tag, bufs, iters, sync_each):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        lib.sbq_kth_value(L.ptr(bufs[i % len(bufs)]), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st)
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    print("%s: %.1f us per call, out=%r nonzero ws=%d" % (tag, (time.perf_counter() - t0) * 1e6 / iters, out[0].item(), int((sw[OLD:] != 0).sum())), flush=True)
loop("same buffer, sync each", [w], 20, True)
loop("same buffer, back to back", [w], 20, False)
loop("4 buffers, sync each", xs, 20, True)
loop("4 buffers, back to back", xs, 20, False)
loop("same buffer, back to back x200", [w], 200, False)

print("---- host time of the call itself")
for knob in (0, 12, 0):
    L.set_tuning(2, knob)
    ts = []
    for i in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lib.sbq_kth_value(L.ptr(w), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append("%.0f+%.0f" % ((t1 - t0) * 1e6, (t2 - t1) * 1e6))
    print("knob2=%d  host call us + sync us: %s" % (knob, " ".join(ts)), flush=True)
L.set_tuning(2, 0)

print("---- loops again, per-call distribution")
def loop2(tag, bufs, iters, sync_each):
    torch.cuda.synchronize()
    ts = []
    t_all = time.perf_counter()
    for i in range(iters):
        t0 = time.perf_counter()
        lib.sbq_kth_value(L.ptr(bufs[i % len(bufs)]), L.BF16, n, 1, n // 2 + 1, L.ptr(out), L.ptr(sw), sw.numel(), st)
        if sync_each: torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) * 1e6
    print("%s: total %.0f us, per-call: %s" % (tag, tot, " ".join("%.0f" % t for t in ts)), flush=True)
loop2("same buffer, sync each", [w], 12, True)
loop2("same buffer, back to back", [w], 12, False)
loop2("4 buffers, sync each", xs, 12, True)
loop2("4 buffers b2b", xs, 12, False)
