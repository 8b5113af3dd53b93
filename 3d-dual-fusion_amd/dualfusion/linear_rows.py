"""torch.nn.Linear whose WEIGHT GRADIENT runs on df3d_rows_grad_weights (round 5, training rows: SURVEY.md section 8f row 4).

The adapter's linear layers (CP/det3d/models/fusion/actr_transformer.py:388-424: the two feed-forward blocks of the dual-query
layer; ms_deform_attn.py:60-64: sampling offsets, attention weights, value and output projections) see 32 k - 240 k rows.  Their
weight gradient grad^T x is a [C_out x rows] . [rows x C_in] product -- a contraction over the ROWS, where the library's fp32
GEMM runs at a fifth of its rate (and, for inputs with a batch dimension, as a batched product plus a sum over the batch).
`Linear` is a drop-in subclass (same parameters, same state_dict keys, same forward values -- the forward IS F.linear); only
the backward differs: grad_input = grad W (library GEMM), grad_weight = df3d_rows_grad_weights_scaled(grad, x) -- fp16 pairs, the
gradient under its own power-of-two block scale, three products, fp32 accumulate: fp32-grade of scale (tests/test_gpu_ops.py),
grad_bias = column sums.  DF3D_LINEAR_WGRAD=0 keeps autograd's own backward (A/B switch, read per call)."""
import os

import torch
from torch import nn

from . import ops as _ops


class _LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        if grad.dtype != torch.float32:               # (a caller that casts the output: the kernels take fp32 rows)
            grad = grad.float()
        g2 = grad.reshape(-1, grad.shape[-1])
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (g2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, x.shape[-1])
            g2c, x2c = g2.contiguous(), x2.contiguous()
            if g2c.shape[1] % 4:                      # (a gate with one output: columns padded to the kernel's 4-channel pieces)
                g2c = torch.nn.functional.pad(g2c, (0, (-g2c.shape[1]) % 4))
            # two-part operands: the gradient under its own power-of-two block scale, the activation at the fixed scale
            gw = _ops.rows_grad_weights(g2c, x2c, x_scale=_ops.rows_pow2_scale(g2c))[:g2.shape[1]]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


def linear(x, weight, bias=None):
    """F.linear with the weight gradient on the row kernel where it applies (CUDA fp32, >= 2048 rows, input channels a multiple
    of 4), else F.linear."""
    # (under torch.autocast -- the reference trains these layers with AMP -- F.linear returns half precision and hands the
    # backward half-precision gradients, which the fp32 row kernels do not take: autocast keeps nn.Linear's own path)
    if (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and not torch.is_autocast_enabled()
            and weight.requires_grad and x.shape[-1] % 4 == 0 and x.numel() // max(x.shape[-1], 1) >= 2048
            and os.environ.get("DF3D_LINEAR_WGRAD", "1") != "0"):
        return _LinearFunction.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class Linear(nn.Linear):
    def forward(self, x):
        return linear(x, self.weight, self.bias)
