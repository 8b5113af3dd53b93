#!/usr/bin/env python3
"""Which BLAS layout is fastest for the two image-side GEMMs of the adapter (fp32, 6 x 150x267 maps)."""
import torch

dev = torch.device("cuda:0")
N, C, H, W = 6, 256, 150, 267
S = H * W
img = torch.randn(N, C, H, W, device=dev)
Wc = torch.randn(144, C, device=dev) * 0.05
Wn = torch.randn(N, 256, 128, device=dev) * 0.05


def bench(name, fn, flops):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    print("%-60s %8.1f us  %6.1f TF  out %s contiguous=%s" % (name, us, flops / us / 1e6, tuple(y.shape),
                                                          y.is_contiguous()))
    return y


f1 = 2.0 * N * S * C * 144
imgf = img.flatten(2)
Wt = Wc.t().contiguous()
u_pm = bench("G1 pixel-major  bmm(img^T view, Wc^T)", lambda: torch.bmm(imgf.transpose(1, 2), Wt.expand(N, C, 144)), f1)
u_pm2 = bench("G1 pixel-major  matmul(img^T view, Wc^T)", lambda: torch.matmul(imgf.transpose(1, 2), Wt), f1)
u_cf = bench("G1 channel-first matmul(Wc, img)", lambda: torch.matmul(Wc, imgf), f1)
u_cv = bench("G1 conv2d 1x1 (MIOpen)", lambda: torch.nn.functional.conv2d(img, Wc[:, :, None, None]), f1)
print("max diff pm vs cf", (u_pm.transpose(1, 2) - u_cf).abs().max().item())
f2 = 2.0 * N * S * 128 * 256
WnT = Wn.transpose(1, 2)
bench("G2 from pixel-major bmm(u[:,:,:128], Wn^T)", lambda: torch.bmm(u_pm[:, :, :128], WnT), f2)
bench("G2 from pixel-major contiguous 128", lambda: torch.bmm(u_pm[:, :, :128].contiguous(), WnT), f2)
bench("G2 from channel-first bmm(u[:, :128]^T, Wn^T)", lambda: torch.bmm(u_cf[:, :128].transpose(1, 2), WnT), f2)
x = torch.randn(31000, 128, device=dev)
W1 = torch.randn(1024, 128, device=dev) * 0.05
b1 = torch.randn(1024, device=dev)
W2 = torch.randn(128, 1024, device=dev) * 0.05
b2 = torch.randn(128, device=dev)
h = bench("FFN linear1 31000x128 -> 1024", lambda: torch.nn.functional.linear(x, W1, b1), 2.0 * 31000 * 128 * 1024)
bench("FFN linear1 + relu_", lambda: torch.relu_(torch.nn.functional.linear(x, W1, b1)), 2.0 * 31000 * 128 * 1024)
try:
    bench("FFN _addmm_activation relu", lambda: torch._addmm_activation(b1, x, W1.t(), use_gelu=False),
          2.0 * 31000 * 128 * 1024)
    print("addmm_act diff", (torch._addmm_activation(b1, x, W1.t()) - torch.relu(torch.nn.functional.linear(x, W1, b1))).abs().max().item())
except Exception as e:  # noqa
    print("no _addmm_activation:", e)
bench("FFN linear2 31000x1024 -> 128", lambda: torch.nn.functional.linear(h, W2, b2), 2.0 * 31000 * 128 * 1024)

# ---- fused split-precision FFN (csrc/ffn.hip) vs the two hipBLASLt GEMMs + ReLU + add + LayerNorm
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
from dualfusion import ops  # noqa: E402
lw, lb = torch.randn(128, device=dev), torch.randn(128, device=dev)
packed = ops.ffn_pack(W1, W2)
fl = 4.0 * 31000 * 128 * 1024
bench("FFN fused kernel (split precision)", lambda: ops.ffn_fused(x, packed, b1, b2, 1024, residual=x, ln_weight=lw,
                                                                  ln_bias=lb), fl)
bench("FFN torch: addmm_act + linear + add + LN", lambda: torch.nn.functional.layer_norm(
    x + torch.nn.functional.linear(torch._addmm_activation(b1, x, W1.t()), W2, b2), (128,), lw, lb), fl)
