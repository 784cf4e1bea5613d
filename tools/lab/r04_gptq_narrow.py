"""Dev tool (round 4): the 16-column strip variant of the GPTQ mat-vec (knob 2 == 22) against the default,
HBM-cold (rotating weight copies), at the LLaMA-7B shapes; checks the result against the default launch."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
stream = torch.cuda.current_stream(dev)

def timed(fn, iters=300, warm=30, rounds=3):
    best = 1e9
    for _ in range(rounds):
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for i in range(iters): fn(i)
        b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best

g = torch.Generator().manual_seed(1)
fns = {4: lib.sbq_vecquant4matmul, 3: lib.sbq_vecquant3matmul, 2: lib.sbq_vecquant2matmul}
bits_list = [int(b) for b in os.environ.get("BITS", "4").split(",")]
for bits in bits_list:
  for batch in (1, 2):
    for in_f, out_f in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (5120, 5120), (8192, 8192)):
        groups = in_f // 128
        H = in_f // 32 * bits
        wb = H * out_f * 4
        copies = max(2, int(3.2e8 // wb) + 1)
        qws = [torch.randint(-2**31, 2**31 - 1, (H, out_f), generator=g, dtype=torch.int64).to(torch.int32).to(dev) for _ in range(copies)]
        sc = (torch.rand(out_f, groups, generator=g) * 0.02 + 0.001).to(dev)
        zr = (torch.rand(out_f, groups, generator=g) * 0.1).to(dev)
        x = torch.randn(batch, in_f, generator=g).to(dev)
        ws = L.fresh_workspace(lib.sbq_gptq_workspace_bytes(batch, in_f, out_f), dev)
        nbytes = wb + 2 * out_f * groups * 4 + (in_f + 2 * out_f) * 4 * batch
        line = "%d-bit B=%d %5d -> %5d (%5.1f MB):" % (bits, batch, in_f, out_f, nbytes / 1e6)
        ref = None
        for k1, k2 in ((0, 0), (0, 25), (0, 23)):
            L.set_tuning(1, k1); L.set_tuning(2, k2)
            y = torch.zeros(batch, out_f, device=dev)
            rc = fns[bits](L.ptr(x), L.ptr(qws[0]), L.ptr(y), L.ptr(sc), L.ptr(zr), batch, in_f, out_f, 128, L.ptr(ws), ws.numel(), st)
            assert rc == 0, rc
            torch.cuda.synchronize()
            if ref is None: ref = y.clone()
            err = ((y - ref).abs().max() / ref.abs().max()).item()
            yy = torch.zeros(batch, out_f, device=dev)
            def run(i):
                fns[bits](L.ptr(x), L.ptr(qws[i % copies]), L.ptr(yy), L.ptr(sc), L.ptr(zr), batch, in_f, out_f, 128, L.ptr(ws), ws.numel(), st)
            t = timed(run)
            line += "  k1=%d,k2=%d: %5.2f us (%.2f) e=%.1e" % (k1, k2, t, nbytes / t / 8e6, err)
        L.set_tuning(1, 0); L.set_tuning(2, 0)
        print(line, flush=True)
        del qws
