"""One QAT step with 4-bit LSQ weights + 4-bit LSQ activations and a 50 % unstructured mask
(BASELINE config 5 in miniature; needs an MI355X).

    python examples/qat_lsq_sparse.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sparsebit_amd.common import Backend  # noqa: E402
from sparsebit_amd.config import quantizer_config, sparser_config  # noqa: E402
from sparsebit_amd.quantizers import build_quantizer  # noqa: E402
from sparsebit_amd.sparsers import build_sparser  # noqa: E402


def main():
    torch.manual_seed(0)
    dev = "cuda"
    conv = torch.nn.Conv2d(64, 128, 3, padding=1).to(dev)
    wq = build_quantizer(quantizer_config("per-channel-symmetric", 4, quantizer="lsq"))
    aq = build_quantizer(quantizer_config("per-tensor-affine", 4, quantizer="lsq", target="feature"))
    for q in (wq, aq):
        q.set_backend(Backend.VIRTUAL)
    x = torch.relu(torch.randn(32, 64, 28, 28, device=dev))
    # calibration: LSQ initialises its step sizes from data, the sparser fixes the mask
    wq.update_observer(conv.weight)
    wq.calc_qparams()
    aq.update_observer(x)
    aq.calc_qparams()
    mask = build_sparser(sparser_config(0.5)).calc_mask(conv.weight)
    wq.enable_quant()
    aq.enable_quant()
    params = [conv.weight, conv.bias, wq.scale, aq.scale]
    opt = torch.optim.SGD(params, lr=1e-3)
    for step in range(3):
        y = torch.nn.functional.conv2d(aq(x), wq(conv.weight * mask), conv.bias, padding=1)
        loss = y.pow(2).mean()
        opt.zero_grad()
        loss.backward()  # STE backward kernel: grads for the weight, the masked-out entries get none
        opt.step()
        print("step %d loss %.5f | mean weight step %.5f, activation step %.5f | sparsity %.3f"
              % (step, loss.item(), wq.scale.abs().mean().item(), aq.scale.abs().item(), 1 - mask.float().mean().item()))
    fused = wq.forward_masked(conv.weight.detach(), mask=mask)  # inference: one fused kernel
    print("fused mask+QDQ equals unfused:", bool(torch.equal(fused, wq(conv.weight * mask).detach())))


if __name__ == "__main__":
    main()
