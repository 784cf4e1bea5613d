"""Generate golden vectors from the REAL reference (authoring container only).

Runs Sparsebit's own CPU implementation of the hot path -- imported from
/root/reference, never copied -- on small seeded inputs and stores inputs and
outputs in tests/golden/ref_golden.npz.  The reference cannot travel to the GPU
box, the vectors can.  Re-run:  HIP_VISIBLE_DEVICES="" python tests/golden/gen_golden.py

Three import stubs are needed for packages absent from this image:
yacs.config.CfgNode, onnx, torchvision.ops.stochastic_depth.
"""
import copy
import math
import os
import sys
import types

os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz")


# ---- import stubs -----------------------------------------------------------------
class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        return copy.deepcopy(self)

    def defrost(self):
        pass

    def freeze(self):
        pass

    def merge_from_file(self, path):
        import yaml

        def merge(dst, src):
            for k, v in src.items():
                if isinstance(v, dict) and isinstance(dst.get(k), dict):
                    merge(dst[k], v)
                else:
                    dst[k] = CfgNode(v) if isinstance(v, dict) else v

        with open(path) as f:
            merge(self, yaml.safe_load(f) or {})

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node = self
            parts = k.split(".")
            for p in parts[:-1]:
                node = node[p]
            node[parts[-1]] = v


def install_stubs():
    yacs = types.ModuleType("yacs")
    yacs_config = types.ModuleType("yacs.config")
    yacs_config.CfgNode = CfgNode
    yacs.config = yacs_config
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_config
    sys.modules["onnx"] = types.ModuleType("onnx")
    tv = types.ModuleType("torchvision")
    tv_ops = types.ModuleType("torchvision.ops")
    tv_sd = types.ModuleType("torchvision.ops.stochastic_depth")
    tv_sd.stochastic_depth = lambda *a, **k: None
    tv.ops = tv_ops
    tv_ops.stochastic_depth = tv_sd
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tv_ops
    sys.modules["torchvision.ops.stochastic_depth"] = tv_sd


def qcfg(qscheme, bit, observer="MINMAX", quantizer="uniform", target_weight=True, layout="NCHW", alpha=1e-3,
         ema_ratio=0.9, aciq="GAUS", pact_alpha=10):
    from sparsebit.quantization.common import QuantTarget

    c = CfgNode()
    c.QSCHEME = qscheme
    c.QUANTIZER = CfgNode({"TYPE": quantizer, "DISABLE": False, "BIT": bit, "PACT": CfgNode({"ALPHA_VALUE": pact_alpha})})
    obs = {"TYPE": observer, "PERCENTILE": CfgNode({"ALPHA": alpha}), "ACIQ": CfgNode({"DISTRIBUTION": aciq})}
    if not target_weight:
        obs["MOVING_AVERAGE"] = CfgNode({"EMA_RATIO": ema_ratio})
    if not target_weight:
        obs["LAYOUT"] = layout
    c.OBSERVER = CfgNode(obs)
    c.TARGET = (QuantTarget.WEIGHT,) if target_weight else (QuantTarget.FEATURE,)
    return c


def main():
    assert not torch.cuda.is_available(), "generate goldens with GPUs hidden (reference CPU path)"
    install_stubs()
    sys.path.insert(0, REF)
    from sparsebit.quantization.quantizers import build_quantizer
    from sparsebit.quantization.common import Backend
    from sparsebit.quantization.quantizers.quant_tensor import MySTE

    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    out = {}
    cases = []

    def weight(shape, spread=True):
        w = torch.randn(*shape, generator=g)
        if spread:
            w = w * torch.logspace(-2, 1, shape[0]).reshape(-1, *([1] * (len(shape) - 1)))
        return w.bfloat16().float()  # bf16-representable values: same tensor feeds every dtype

    # ---- uniform quantizer end to end: observer -> qparams -> forward ---------------
    def run_quantizer(name, x_list, cfg, backend=Backend.VIRTUAL):
        q = build_quantizer(cfg)
        q.set_backend(backend)
        for x in x_list:
            q.update_observer(x)
        scale, zp = q.calc_qparams()
        q.enable_quant()
        dq = q(x_list[0])
        out[name + "/x"] = x_list[0].numpy()
        for i, x in enumerate(x_list[1:]):
            out[name + "/x%d" % (i + 1)] = x.numpy()
        out[name + "/scale"] = scale.detach().numpy().reshape(-1)
        out[name + "/zero_point"] = zp.detach().numpy().reshape(-1)
        out[name + "/min_val"] = q.observer.min_val.numpy().reshape(-1)
        out[name + "/max_val"] = q.observer.max_val.numpy().reshape(-1)
        out[name + "/dq"] = dq.detach().numpy()
        out[name + "/meta"] = np.array([q.qdesc.qmin, q.qdesc.qmax, q.qdesc.ch_axis,
                                        int(q.qdesc.is_perchannel), int(q.qdesc.is_symmetric)], dtype=np.int64)
        cases.append(name)
        return q

    w_lin = weight((64, 96))
    w_conv = weight((32, 16, 3, 3))
    w_rag = weight((24, 3, 7, 7))  # inner = 147: not a multiple of 8
    w_zero = weight((16, 40))
    w_zero[3] = 0.0  # zero row -> scale floor 1e-6
    a_nchw = [torch.randn(4, 8, 6, 6, generator=g).bfloat16().float() for _ in range(2)]
    a_nlc = [torch.randn(3, 17, 48, generator=g).bfloat16().float() for _ in range(2)]
    a_relu = [torch.relu(a) for a in a_nchw]

    for scheme in ("per-channel-symmetric", "per-channel-affine", "per-tensor-symmetric", "per-tensor-affine"):
        for bit in (8, 4):
            for wname, w in (("lin", w_lin), ("conv", w_conv), ("rag", w_rag), ("zero", w_zero)):
                run_quantizer("uni/%s/%d/%s" % (scheme, bit, wname), [w], qcfg(scheme, bit))
    # TensorRT backend (zp must be 0): symmetric only
    run_quantizer("trt/per-channel-symmetric/8/lin", [w_lin], qcfg("per-channel-symmetric", 8), Backend.TENSORRT)
    run_quantizer("trt/per-tensor-symmetric/8/nchw", a_nchw, qcfg("per-tensor-symmetric", 8, target_weight=False),
                  Backend.TENSORRT)
    # activations, per tensor (every shipped config), two cached batches
    for scheme in ("per-tensor-symmetric", "per-tensor-affine"):
        run_quantizer("act/%s/8/nchw" % scheme, a_nchw, qcfg(scheme, 8, target_weight=False, layout="NCHW"))
        run_quantizer("act/%s/8/nlc" % scheme, a_nlc, qcfg(scheme, 8, target_weight=False, layout="NLC"))
        run_quantizer("act/%s/8/relu" % scheme, a_relu, qcfg(scheme, 8, target_weight=False, layout="NCHW"))
    # per-channel activation, single batch (multi-batch is reference quirk Q3)
    run_quantizer("act/per-channel-affine/8/nchw1", a_nchw[:1], qcfg("per-channel-affine", 8, target_weight=False))
    run_quantizer("act/per-channel-symmetric/8/nlc1", a_nlc[:1],
                  qcfg("per-channel-symmetric", 8, target_weight=False, layout="NLC"))

    # ---- percentile observer -----------------------------------------------------------
    for scheme in ("per-channel-symmetric", "per-channel-affine", "per-tensor-symmetric", "per-tensor-affine"):
        for alpha in (1e-3, 0.05):
            run_quantizer("pct/%s/%g/lin" % (scheme, alpha), [w_lin], qcfg(scheme, 8, "PERCENTILE", alpha=alpha))
            run_quantizer("pct/%s/%g/conv" % (scheme, alpha), [w_conv], qcfg(scheme, 8, "PERCENTILE", alpha=alpha))
    run_quantizer("pct/per-tensor-affine/0.01/nlc", a_nlc,
                  qcfg("per-tensor-affine", 8, "PERCENTILE", target_weight=False, layout="NLC", alpha=0.01))
    run_quantizer("pct/per-tensor-symmetric/0.01/relu", a_relu,
                  qcfg("per-tensor-symmetric", 8, "PERCENTILE", target_weight=False, alpha=0.01))

    # ---- MSE observer: per tensor works on the reference CPU path; per channel only
    # when inner == C, and then silently wrong -> not generated ---------
    for scheme in ("per-tensor-symmetric", "per-tensor-affine"):
        for bit in (8, 4):
            run_quantizer("mse/%s/%d/lin" % (scheme, bit), [w_lin], qcfg(scheme, bit, "MSE"))
            run_quantizer("mse/%s/%d/nchw" % (scheme, bit), a_nchw, qcfg(scheme, bit, "MSE", target_weight=False))

    # ---- LSQ init (lsq.py:32-51) and forward ---------------------------------------------
    for scheme in ("per-channel-symmetric", "per-tensor-symmetric", "per-channel-affine"):
        for bit in (4, 8):
            run_quantizer("lsq/%s/%d/conv" % (scheme, bit), [w_conv], qcfg(scheme, bit, quantizer="lsq"))
    run_quantizer("lsq/per-tensor-affine/4/relu", a_relu[:1],
                  qcfg("per-tensor-affine", 4, quantizer="lsq", target_weight=False))

    # ---- hand KATs through the reference's own fake-quant (quant_tensor.py:159-185) -------
    from sparsebit.quantization.quantizers.quant_tensor import ort_fake_quant
    from sparsebit.quantization.quantizers.quant_descriptor import QuantDescriptor

    kat_x = torch.tensor([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.5, 200, -128.5, -129.5, 0.0, -0.0,
                          1e-30, -1e-30, 3e38, -3e38, float("inf"), float("-inf")])
    d8s = QuantDescriptor(qcfg("per-tensor-symmetric", 8))
    d8a = QuantDescriptor(qcfg("per-tensor-affine", 8))
    out["kat/x"] = kat_x.numpy()
    out["kat/int8_s1_zp0"] = ort_fake_quant(kat_x, torch.tensor([1.0]), torch.tensor([0.0]), d8s).numpy()
    out["kat/uint8_s1_zp3.5"] = ort_fake_quant(kat_x, torch.tensor([1.0]), torch.tensor([3.5]), d8a).numpy()
    out["kat/uint8_s0.3_zp2.5"] = ort_fake_quant(kat_x, torch.tensor([0.3]), torch.tensor([2.5]), d8a).numpy()
    # dense tie test: x = (k + 0.5) * s for an "awkward" s
    s_tie = torch.tensor([0.0123])
    kk = torch.arange(-140, 140, dtype=torch.float32)
    x_tie = ((kk + 0.5) * s_tie).bfloat16().float()
    out["kat/tie_x"] = x_tie.numpy()
    out["kat/tie_s"] = s_tie.numpy()
    out["kat/tie_dq"] = ort_fake_quant(x_tie, s_tie, torch.tensor([0.0]), d8s).numpy()

    # ---- unstructured L1 mask (sparse/sparsers/l1norm.py) ----------------------------------
    from sparsebit.sparse.sparsers import build_sparser

    for ratio in (0.0, 0.3, 0.5, 0.9, 1.0):
        sc = CfgNode({"SPARSER": CfgNode({"TYPE": "unstructed", "STRATEGY": "l1norm", "RATIO": ratio})})
        sp = build_sparser(sc, opr=None)
        for wname, w in (("lin", w_lin), ("conv", w_conv)):
            m = sp.calc_mask(w)
            out["mask/%g/%s" % (ratio, wname)] = m.numpy().astype(np.uint8)
    wk = torch.tensor([[1.0, -2, 2, 3], [0, -3, 4, 2]])
    sc = CfgNode({"SPARSER": CfgNode({"TYPE": "unstructed", "STRATEGY": "l1norm", "RATIO": 0.5})})
    out["mask/kat_x"] = wk.numpy()
    out["mask/kat"] = build_sparser(sc, opr=None).calc_mask(wk).numpy().astype(np.uint8)
    # mask * weight then LSQ 4-bit weight quantizer (config 5's functional composition)
    sp = build_sparser(sc, opr=None)
    m = sp.calc_mask(w_conv)
    q = build_quantizer(qcfg("per-channel-symmetric", 4, quantizer="lsq"))
    q.set_backend(Backend.VIRTUAL)
    q.update_observer(w_conv)
    q.calc_qparams()
    q.enable_quant()
    out["maskq/x"] = w_conv.numpy()
    out["maskq/mask"] = m.numpy().astype(np.uint8)
    out["maskq/scale"] = q.scale.detach().numpy().reshape(-1)
    out["maskq/dq"] = q(w_conv * m).detach().numpy()

    # ---- STE backward: MySTE.backward (quant_tensor.py:45-71), elementwise gs/gz reduced
    # here to the parameter shape like the CUDA kernels do ---------------------------------
    class _Ctx:
        pass

    def ste_bwd(name, x, scale, zp, qdesc, ch_axis):
        ctx = _Ctx()
        s = scale.clone().requires_grad_(True)
        z = zp.round().clone().requires_grad_(True)
        ctx.saved_tensors = (x, s, z)
        ctx.qdesc = qdesc
        gout = torch.randn(x.shape, generator=g)
        with torch.no_grad():
            gin, gs, gz, _, _ = MySTE.backward(ctx, gout)
        dims = [d for d in range(x.dim()) if not (scale.numel() > 1 and d == ch_axis)]
        out[name + "/x"] = x.numpy()
        out[name + "/gy"] = gout.numpy()
        out[name + "/scale"] = scale.numpy().reshape(-1)
        out[name + "/zero_point"] = zp.numpy().reshape(-1)
        out[name + "/gx"] = gin.numpy()
        out[name + "/gs"] = gs.double().sum(dim=dims).float().numpy().reshape(-1)
        out[name + "/gzp"] = gz.double().sum(dim=dims).float().numpy().reshape(-1)
        out[name + "/meta"] = np.array([qdesc.qmin, qdesc.qmax, ch_axis], dtype=np.int64)

    d4 = QuantDescriptor(qcfg("per-channel-symmetric", 4))
    s4 = (2 * w_conv.abs().mean(dim=(1, 2, 3)) / math.sqrt(7)).reshape(-1, 1, 1, 1)
    ste_bwd("bwd/pc4", w_conv, s4, torch.zeros_like(s4), d4, 0)
    d8a_t = QuantDescriptor(qcfg("per-tensor-affine", 8, target_weight=False))
    ste_bwd("bwd/pt8a", a_nchw[0], torch.tensor([0.02]), torch.tensor([117.0]), d8a_t, 1)
    d8a_c = QuantDescriptor(qcfg("per-channel-affine", 8, target_weight=False))
    sc8 = torch.linspace(0.01, 0.03, 8).reshape(1, 8, 1, 1)
    ste_bwd("bwd/pc8a_nchw", a_nchw[0], sc8, torch.full_like(sc8, 100.0), d8a_c, 1)

    # ---- GPTQ 4-bit: find_params / quantize / pack from the reference's llama utils, with
    # its CUDA loader stubbed; expected output = quantized nn.Linear (test_cuda_kernel.py) ----
    llama = os.path.join(REF, "large_language_models/llama/quantization")
    sys.path.insert(0, llama)
    lk = types.ModuleType("utils.load_cuda_kernel")
    lk.cuda_kernel = None
    import importlib

    utils_pkg = importlib.import_module("utils")
    sys.modules["utils.load_cuda_kernel"] = lk
    rq = importlib.import_module("utils.quant")
    for name, (B, M, N, GS) in {"g128": (3, 256, 72, 128), "g-1": (2, 136, 40, -1), "rag": (5, 384, 33, 128)}.items():
        torch.manual_seed(7)
        layer = torch.nn.Linear(M, N)
        vec = torch.randn(B, M, generator=g)
        qz = rq.Quantizer()
        qz.configure(bit=4, perchannel=True, sym=False, mse=False)
        qz.find_params(layer.weight.data, weight=True, groupsize=GS)
        w_orig = layer.weight.data.clone()
        layer.weight.data = rq.quantize(layer.weight.data.view(-1, M if GS == -1 else GS), qz.scale.view(-1, 1),
                                        qz.zero.view(-1, 1), qz.maxq).view(N, M)
        ql = rq.QuantLinear(M, N, bit=4, groupsize=GS)
        ql.pack(layer, qz.scale, qz.zero)
        with torch.no_grad():
            y = layer(vec)
        out["gptq/%s/w" % name] = w_orig.numpy()
        out["gptq/%s/wq" % name] = layer.weight.data.numpy()
        out["gptq/%s/x" % name] = vec.numpy()
        out["gptq/%s/scale" % name] = qz.scale.numpy().reshape(N, -1)
        out["gptq/%s/zero" % name] = qz.zero.numpy().reshape(N, -1)
        out["gptq/%s/qweight" % name] = ql.qweight.numpy()
        out["gptq/%s/scales" % name] = ql.scales.numpy().reshape(N, -1)
        out["gptq/%s/zeros" % name] = ql.zeros.numpy().reshape(N, -1)
        out["gptq/%s/bias" % name] = ql.bias.detach().numpy()
        out["gptq/%s/y" % name] = y.numpy()
        out["gptq/%s/meta" % name] = np.array([B, M, N, GS], dtype=np.int64)

    # ---- widened set: remaining observers / quantizers on the same kernels.
    # Appended last so that every earlier vector keeps its value. ---------------------------
    cases2 = []

    def run2(name, x_list, cfg, backend=Backend.VIRTUAL, forward_x=None):
        n0 = len(cases)
        q = run_quantizer(name, x_list, cfg, backend)
        cases.pop()  # keep the original `cases` list (and the tests parametrised on it) unchanged
        assert len(cases) == n0
        cases2.append(name)
        return q

    for scheme in ("per-tensor-symmetric", "per-tensor-affine"):
        run2("ma/%s/8/nchw" % scheme, a_nchw, qcfg(scheme, 8, "MOVING_AVERAGE", target_weight=False, ema_ratio=0.9))
        run2("ma/%s/8/nlc" % scheme, a_nlc, qcfg(scheme, 8, "MOVING_AVERAGE", target_weight=False, layout="NLC",
                                               ema_ratio=0.7))
    for dist_ in ("GAUS", "LAPLACE"):
        for scheme in ("per-channel-symmetric", "per-tensor-symmetric", "per-tensor-affine", "per-channel-affine"):
            for bit in (8, 4):
                run2("aciq/%s/%s/%d/lin" % (dist_, scheme, bit), [w_lin], qcfg(scheme, bit, "ACIQ", aciq=dist_))
        for scheme in ("per-tensor-symmetric", "per-tensor-affine"):
            run2("aciq/%s/%s/8/nchw" % (dist_, scheme), a_nchw, qcfg(scheme, 8, "ACIQ", target_weight=False, aciq=dist_))
            run2("aciq/%s/%s/8/relu" % (dist_, scheme), a_relu, qcfg(scheme, 8, "ACIQ", target_weight=False, aciq=dist_))
    for scheme in ("per-tensor-symmetric", "per-tensor-affine"):
        for bit in (8, 4):
            run2("pact/%s/%d/nchw" % (scheme, bit), a_nchw, qcfg(scheme, bit, quantizer="pact", target_weight=False,
                                                             pact_alpha=1.5))
    run2("pact/per-tensor-affine/4/relu", a_relu, qcfg("per-tensor-affine", 4, quantizer="pact", target_weight=False,
                                                      pact_alpha=2.0))
    for scheme in ("per-channel-symmetric", "per-tensor-symmetric", "per-tensor-affine"):
        run2("dorefa/%s/4/conv" % scheme, [w_conv], qcfg(scheme, 4, quantizer="dorefa"))
    for bit in (4, 8):
        run2("lsqp/per-channel-symmetric/%d/conv" % bit, [w_conv], qcfg("per-channel-symmetric", bit, quantizer="lsq+"))
        run2("lsqp/per-tensor-affine/%d/nchw" % bit, a_nchw, qcfg("per-tensor-affine", bit, quantizer="lsq+",
                                                                target_weight=False))
        run2("lsqp/per-tensor-affine/%d/relu" % bit, a_relu, qcfg("per-tensor-affine", bit, quantizer="lsq+",
                                                                target_weight=False))
    # ---- GPTQ 3- and 2-bit, same recipe as the 4-bit block above.
    # Appended after everything else: no earlier vector changes. -------------------------------
    g23 = torch.Generator().manual_seed(23)
    gptq23 = {"b3/g128": (3, 3, 256, 72, 128), "b3/g-1": (3, 2, 136, 40, -1), "b3/rag": (3, 5, 384, 33, 128),
              "b3/strip": (3, 1, 512, 64, 256),
              "b2/g64": (2, 3, 256, 72, 64), "b2/g-1": (2, 2, 136, 40, -1), "b2/rag": (2, 5, 384, 33, 128),
              "b2/strip": (2, 2, 512, 64, 128)}
    for name, (bit, B, M, N, GS) in gptq23.items():
        torch.manual_seed(11)
        layer = torch.nn.Linear(M, N)
        vec = torch.randn(B, M, generator=g23)
        qz = rq.Quantizer()
        qz.configure(bit=bit, perchannel=True, sym=False, mse=False)
        qz.find_params(layer.weight.data, weight=True, groupsize=GS)
        w_orig = layer.weight.data.clone()
        layer.weight.data = rq.quantize(layer.weight.data.view(-1, M if GS == -1 else GS), qz.scale.view(-1, 1),
                                        qz.zero.view(-1, 1), qz.maxq).view(N, M)
        ql = rq.QuantLinear(M, N, bit=bit, groupsize=GS)
        ql.pack(layer, qz.scale, qz.zero)
        with torch.no_grad():
            y = layer(vec)
        out["gptq/%s/w" % name] = w_orig.numpy()
        out["gptq/%s/wq" % name] = layer.weight.data.numpy()
        out["gptq/%s/x" % name] = vec.numpy()
        out["gptq/%s/scale" % name] = qz.scale.numpy().reshape(N, -1)
        out["gptq/%s/zero" % name] = qz.zero.numpy().reshape(N, -1)
        out["gptq/%s/qweight" % name] = ql.qweight.numpy()
        out["gptq/%s/scales" % name] = ql.scales.numpy().reshape(N, -1)
        out["gptq/%s/zeros" % name] = ql.zeros.numpy().reshape(N, -1)
        out["gptq/%s/bias" % name] = ql.bias.detach().numpy()
        out["gptq/%s/y" % name] = y.numpy()
        out["gptq/%s/meta" % name] = np.array([B, M, N, GS, bit], dtype=np.int64)
    out["gptq23"] = np.array(sorted(gptq23))

    # ---- pack32 -> pack8 checkpoint conversion: run the reference's script on a tiny checkpoint
    import tempfile

    qlora = os.path.join(REF, "large_language_models/alpaca-qlora")
    sys.path.insert(0, qlora)
    conv = importlib.import_module("convert_pack32topack8")
    with tempfile.TemporaryDirectory() as td:
        src, dst = os.path.join(td, "a.pth"), os.path.join(td, "b.pth")
        qw32 = torch.from_numpy(out["gptq/g128/qweight"]).clone()
        torch.save({"model": {"layers.0.qweight": qw32, "layers.0.scales": torch.ones(3)}}, src)
        conv.main(types.SimpleNamespace(checkpoint=src, output=dst))
        got = torch.load(dst)["model"]
    out["pack8/qweight32"] = qw32.numpy()
    out["pack8/qweight8"] = got["layers.0.qweight"].numpy()
    out["cases2"] = np.array(cases2)
    out["cases"] = np.array(cases)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays,", len(cases), "quantizer cases,",
          os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
