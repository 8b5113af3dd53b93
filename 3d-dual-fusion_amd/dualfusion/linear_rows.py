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


class _LinearBf16Function(torch.autograd.Function):
    """Linear layer of the bf16 mixed-precision training mode (BASELINE configs[2] / [3]): operands rounded to bfloat16 (the
    result stays bfloat16 for the next layer of the feed-forward branch), fp32 accumulation in the library's products.  The
    weight gradient grad^T x contracts over the ROWS (131 k at the TransFusion shape): as ONE product the library runs it at a
    few per cent of its rate (442 us for 128 x 1024 outputs), so it is taken per sample -- a batched product over the leading
    dimension (or over 16 row chunks) -- and the partial results are summed in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x16, w16 = x.to(torch.bfloat16), weight.to(torch.bfloat16)
        ctx.save_for_backward(x16, w16)
        ctx.xdtype, ctx.has_bias = x.dtype, bias is not None
        return torch.nn.functional.linear(x16, w16, None if bias is None else bias.to(torch.bfloat16))

    @staticmethod
    def backward(ctx, grad):
        x16, w16 = ctx.saved_tensors
        g16 = grad.to(torch.bfloat16)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = (g16 @ w16).to(ctx.xdtype)
        if ctx.needs_input_grad[1]:
            g3, x3 = g16, x16
            if g3.dim() != 3:
                g2, x2 = g3.reshape(-1, g3.shape[-1]), x3.reshape(-1, x3.shape[-1])
                parts = 16 if g2.shape[0] % 16 == 0 and g2.shape[0] >= 16 * 1024 else 1
                g3, x3 = g2.view(parts, -1, g2.shape[1]), x2.view(parts, -1, x2.shape[1])
            gw = torch.bmm(g3.transpose(1, 2), x3).sum(0, dtype=torch.float32)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g16.reshape(-1, g16.shape[-1]).sum(0, dtype=torch.float32)
        return gx, gw, gb


def linear_bf16(x, weight, bias=None):
    """x W^T + b with bfloat16 operands and a bfloat16 result, differentiable (fp32 parameter gradients)."""
    return _LinearBf16Function.apply(x, weight, bias)


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
