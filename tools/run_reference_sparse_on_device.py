"""Run the REAL reference's SparseModel on the MI355X with sparsebit_amd.plugin.install() and compare its masks with the
reference ALONE on the host (its torch.sort path, GPUs hidden, no plugin) in a subprocess of this very script.

    python tools/run_reference_sparse_on_device.py [--reference /path/to/Sparsebit] > profiles/r06_reference_sparsemodel_on_device.log

The reference tree is not part of this repository (tools/run_reference_on_device.py: same conventions, same import stubs).
What is run, in both processes from the same seed: SparseModel(net, SPARSER: unstructed / l1norm / 0.5) of a small
convolutional network (conv 7x7 stem, six 3x3 / 1x1 convolutions, a linear layer -- the layer kinds of
sparsebit/sparse/modules), `calc_params()` (sparse/sparse_model.py:107-113), a forward.  Compared: every layer's w_mask
element for element, the logits.  Timed on the device: calc_params with the model-wide route (plugin.route_sparse_params:
ONE grouped selection in front of the reference's per-layer loop) and with the per-layer sparsers alone.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import run_reference_on_device as R  # noqa: E402

YAML = "SPARSER:\n  TYPE: %s\n  STRATEGY: l1norm\n  RATIO: %s\n" % (os.environ.get("SBQ_SPARSE_TYPE", "unstructed"),
                                                                         os.environ.get("SBQ_SPARSE_RATIO", "0.5"))


def make_net():
    import torch
    import torch.nn as nn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
            chans = [(64, 64, 3), (64, 128, 1), (128, 128, 3), (128, 256, 1), (256, 256, 3), (256, 512, 3)]
            self.convs = nn.ModuleList(nn.Conv2d(a, b, k, padding=k // 2, bias=True) for a, b, k in chans)
            self.pool = nn.AdaptiveAvgPool2d(1)
            self.fc = nn.Linear(512, 100)

        def forward(self, x):
            x = torch.relu(self.stem(x))
            x = torch.relu(self.convs[0](x))
            x = torch.relu(self.convs[1](x))
            x = torch.relu(self.convs[2](x))
            x = torch.relu(self.convs[3](x))
            x = torch.relu(self.convs[4](x))
            x = torch.relu(self.convs[5](x))
            return self.fc(torch.flatten(self.pool(x), 1))

    torch.manual_seed(0)
    return Net().eval()


def build(device):
    import contextlib
    import io

    import torch
    from sparsebit.sparse import SparseModel, parse_sconfig

    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        f.write(YAML)
    cfg = parse_sconfig(f.name)
    os.unlink(f.name)
    cfg.defrost() if hasattr(cfg, "defrost") else None
    cfg.DEVICE = device
    with contextlib.redirect_stdout(io.StringIO()):  # (the reference prints the traced graph)
        sm = SparseModel(make_net().to(device), cfg)
    return sm.to(device)


def masks_of(sm):
    return {n: m.w_mask.detach().cpu() for n, m in sm.model.named_modules() if hasattr(m, "w_mask")}


def host_side(ref, out_path):
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    R.setup(ref)
    import torch

    sm = build("cpu")
    t0 = time.perf_counter()
    sm.calc_params()
    dt = time.perf_counter() - t0
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 64, 64, generator=g)
    with torch.no_grad():
        y = sm(x)
    torch.save({"masks": masks_of(sm), "y": y, "calc_params_ms": dt * 1e3}, out_path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=None)
    ap.add_argument("--host-side", default=None)
    args = ap.parse_args()
    ref = R.find_reference(args.reference)
    if args.host_side:
        return host_side(ref, args.host_side)
    with tempfile.TemporaryDirectory() as tmp:
        host_out = os.path.join(tmp, "host.pt")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference", ref, "--host-side", host_out],
                           capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise SystemExit("host side failed:\n" + r.stderr[-3000:])
        R.setup(ref)
        import torch

        from sparsebit_amd import plugin

        plugin.preinstall()
        import sparsebit  # noqa: F401

        installed = plugin.install()
        host = torch.load(host_out)
        sm = build("cuda")
        sparsers = [m.sparser for m in sm.model.modules() if getattr(m, "sparser", None) is not None]
        print("reference: %s" % ref)
        print("installed sparsers: %r; SparseModel sparsers: %d, all sparsebit_amd: %s" % (
            installed["sparsers"], len(sparsers), all(type(s).__module__.startswith("sparsebit_amd") for s in sparsers)))
        sm.calc_params()
        torch.cuda.synchronize()
        dev_masks = masks_of(sm)
        same = {n: bool(torch.equal(dev_masks[n], host["masks"][n])) and dev_masks[n].dtype == host["masks"][n].dtype
                for n in host["masks"]}
        kept = {n: float(dev_masks[n].float().mean()) for n in dev_masks}
        print("w_mask of every layer == the reference alone on the host (torch.sort), element for element: %s" % all(same.values()))
        for n in same:
            print("   %-12s %-18s kept %.4f  %s" % (n, tuple(dev_masks[n].shape), kept[n], "==" if same[n] else "DIFFERS"))
        g = torch.Generator().manual_seed(1)
        x = torch.randn(4, 3, 64, 64, generator=g).cuda()
        with torch.no_grad():
            y = sm(x).cpu()
        print("logits: max |device - host| = %.3e (max |host| %.3e): the masks are identical, the convolutions are MIOpen's "
              "on the device and oneDNN's on the host" % (float((y - host["y"]).abs().max()), float(host["y"].abs().max())))

        def timed(fn, iters=30):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3

        routed = timed(sm.calc_params)
        orig = type(sm).calc_params

        def unrouted():
            pre = None
            for node in sm.model.graph.nodes:  # the reference's loop, verbatim in behaviour (sparse_model.py:107-113)
                if node.op == "call_module":
                    m = getattr(sm.model, node.target, None)
                    if getattr(m, "sparser", None):
                        pre = m.calc_mask(pre)

        per_layer = timed(unrouted)
        print("calc_params on the device: %.3f ms with the model-wide route (one grouped selection + %d mask passes), %.3f ms "
              "layer by layer through the installed sparsers; the reference alone on the host: %.1f ms"
              % (routed, len(sparsers), per_layer, host["calc_params_ms"]))
        del orig
        print(json.dumps({"all_masks_equal": all(same.values()), "layers": len(same), "routed_ms": round(routed, 3),
                          "per_layer_ms": round(per_layer, 3), "host_reference_ms": round(host["calc_params_ms"], 1)}))


if __name__ == "__main__":
    main()
