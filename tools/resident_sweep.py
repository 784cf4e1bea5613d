"""Dev tool: pipelined vs resident schedule of the forward QDQ over tensor sizes (rows x 4096), C ABI loop,
rotating buffers (> 256 MiB).  knob 3: 1 = pipelined only, 0 = the library's choice, 2 = resident forced."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
st = L.stream_ptr(dev)
def timed(fn, iters=300, warm=30):
    best = 1e9
    for _ in range(3):
        for i in range(warm): fn(i)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters): fn(i)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / iters)
    return best
w = torch.randn(8192, 8192, device=dev).bfloat16(); o = torch.empty_like(w)
s = torch.ones(8192, device=dev); z = torch.zeros(8192, device=dev)
for _ in range(2000):
    lib.sbq_quant_perchannel_forward(L.ptr(w), 2, L.ptr(o), 2, None, 0, L.ptr(s), L.ptr(z), 1, 8192, 8192, -128, 127, 0, st)
torch.cuda.synchronize()
del w, o
rows_list = [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2048, 2560, 3072, 3584, 4096, 4608, 5120, 6144, 8192]
for dt, did, odt, odid, esz, osz in ((torch.bfloat16, 2, torch.bfloat16, 2, 2, 2), (torch.float32, 0, torch.float32, 0, 4, 4),
                                     (torch.bfloat16, 2, torch.float32, 0, 2, 4)):
    for rows in rows_list:
        n = rows * 4096
        nbuf = max(2, min(16, int(8e8 // (n * (esz + osz))) + 1))
        xs = [(torch.randn(rows, 4096, device=dev) * torch.logspace(-2, 1, rows, device=dev).unsqueeze(1)).to(dt) for _ in range(nbuf)]
        ys = [torch.empty(rows, 4096, device=dev, dtype=odt) for _ in range(nbuf)]
        sc = xs[0].float().abs().amax(1) * 2 / 255; zp = torch.zeros(rows, device=dev)
        def run(i):
            j = i % nbuf
            rc = lib.sbq_quant_perchannel_forward(L.ptr(xs[j]), did, L.ptr(ys[j]), odid, None, 0, L.ptr(sc), L.ptr(zp), 1, rows, 4096, -128, 127, 0, st)
            assert rc == 0
        line = "%s->%s %5d x 4096 (%6d slabs):" % (str(dt)[6:], str(odt)[6:], rows, n // 2048)
        for k in (1, 0, 2):
            L.set_tuning(3, k)
            t = timed(run)
            line += "  knob3=%d %7.2f us %5.2f TB/s" % (k, t, n * (esz + osz) / t / 1e6)
        L.set_tuning(3, 0)
        print(line, flush=True)
        del xs, ys
