"""Dev tool: all conv / fc weights of a ResNet-50 (synthetic, fp32 masters) quantized per layer vs in one launch."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd import ops

def resnet50_shapes():
    shapes = [(64, 3, 7, 7)]
    inp = 64
    for width, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            shapes += [(width, inp, 1, 1), (width, width, 3, 3), (width * 4, width, 1, 1)]
            if b == 0:
                shapes.append((width * 4, inp, 1, 1))
            inp = width * 4
    shapes.append((1000, 2048))
    return shapes

def deit_small_shapes():
    s = [(384, 3, 16, 16)]
    for _ in range(12):
        s += [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
    s.append((1000, 384))
    return s

def timed(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, (time.perf_counter() - t0) * 1e6 / iters

for name, shapes, dtype in (("resnet50", resnet50_shapes(), torch.float32), ("resnet50", resnet50_shapes(), torch.bfloat16),
                            ("deit-small", deit_small_shapes(), torch.float32)):
    g = torch.Generator().manual_seed(0)
    ws = [torch.randn(s, generator=g).to(dtype).cuda() for s in shapes]
    entries = []
    for w in ws:
        mn, mx, _ = ops.channel_stats(w, 0, True)
        s, z = ops.qparams_from_minmax(mn, mx, -8, 7, True)
        entries.append((w, s, z, -8, 7))
    ok = [e for e in entries if ops.GroupFakeQuant.supports(e[0])]
    rest = [e for e in entries if not ops.GroupFakeQuant.supports(e[0])]
    n_elem = sum(e[0].numel() for e in entries)
    esz = ws[0].element_size()
    outs = [torch.empty_like(e[0]) for e in entries]
    def per_layer():
        for (w, s, z, lo, hi) in entries:
            ops.fake_quant(w, s, z, lo, hi, 0, out_dtype=dtype)
    gq = ops.GroupFakeQuant(ok, out_dtype=dtype)
    def grouped():
        gq()
        for (w, s, z, lo, hi) in rest:
            ops.fake_quant(w, s, z, lo, hi, 0, out_dtype=dtype)
    for (w, s, z, lo, hi), y in zip(ok, gq()):
        assert torch.equal(y, ops.fake_quant(w, s, z, lo, hi, 0, out_dtype=dtype))
    t1 = timed(per_layer); t2 = timed(grouped)
    print("%s %s: %d tensors (%d grouped, %d tiles), %.1f M elements, %.0f MB traffic" % (name, str(dtype)[6:], len(entries), len(ok), gq.n_tiles, n_elem / 1e6, 2 * esz * n_elem / 1e6))
    print("   per layer : %8.1f us GPU  %8.1f us wall" % t1)
    print("   one launch: %8.1f us GPU  %8.1f us wall   (%.2f TB/s)" % (t2[0], t2[1], 2 * esz * n_elem / t2[0] / 1e6))
