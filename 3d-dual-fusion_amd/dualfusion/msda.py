"""MSDeformAttn module + autograd Function shell with the reference's names and signatures
(CP/det3d/models/model_utils/ops/modules/ms_deform_attn.py:33-190,
 .../ops/functions/ms_deform_attn_func.py:21-38).  The sampling kernel is csrc/msda.hip."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.nn.init import constant_, xavier_uniform_

from . import ops as _ops
from ._lib import Df3dError


class MSDeformAttnFunction(Function):
    """forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
    im2col_step) -> [N, Lq, M*D].  im2col_step is accepted for signature parity; the HIP kernels
    need no batch chunking.  backward -> (grad_value, None, None, grad_sampling_loc, grad_attn_weight, None) like
    ms_deform_attn_func.py:31-38 (csrc/msda.hip, df3d_ms_deform_attn_backward)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        value, sampling_locations = value.contiguous(), sampling_locations.contiguous()
        value_spatial_shapes, value_level_start_index = value_spatial_shapes.contiguous(), value_level_start_index.contiguous()
        attention_weights = attention_weights.contiguous()
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return _ops.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                           attention_weights)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lstart, loc, aw = ctx.saved_tensors
        gv, gl, ga = _ops.ms_deform_attn_backward(value, shapes, lstart, loc, aw, grad_output.contiguous())
        return gv, None, None, gl, ga, None


def _is_power_of_2(n):
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, q_model=256, n_levels=4, n_heads=8, n_points=4, q_method=None, q_rep_place=None):
        """`q_model` is accepted and ignored exactly like the reference (ms_deform_attn.py:34,65-68)."""
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("d_model // n_heads should be a power of 2 for the vectorised sampling kernel")
        self.im2col_step = 64
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self.q_method = q_method
        self.q_rep_place = q_rep_place
        if q_method == 'gating':
            from .actr import attn_dict
            self.q_gating = attn_dict['BiGateSum1D_2'](d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.n_heads, 1, 1, 2).repeat(
            1, self.n_levels, self.n_points, 1)
        for i in range(self.n_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid_init.view(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def project_value(self, input_flatten, input_padding_mask=None):
        N, Len_in, _ = input_flatten.shape
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        return value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, i_query=None):
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        assert input_spatial_shapes.shape[0] == self.n_levels
        value = self.project_value(input_flatten, input_padding_mask)
        weight_query = query
        if self.q_method is not None:
            assert i_query is not None
            assert self.q_rep_place is not None
            if self.q_method == 'gating':
                g_query, g_i_query = self.q_gating(query, i_query)
                new_query = g_query + g_i_query - query - i_query
            elif self.q_method == 'sum':
                new_query = query + i_query
            elif self.q_method == 'image':
                new_query = i_query
            else:
                raise NotImplementedError('q_method must be among ["gating", "sum", "image"]')
            if 'offset' in self.q_rep_place:
                query = new_query
            if 'weight' in self.q_rep_place:
                weight_query = new_query
        sampling_offsets = self.sampling_offsets(query).view(N, Len_q, self.n_heads, self.n_levels, self.n_points, 2)
        attention_weights = self.attention_weights(weight_query).view(N, Len_q, self.n_heads,
                                                                      self.n_levels * self.n_points)
        attention_weights = F.softmax(attention_weights, -1).view(N, Len_q, self.n_heads, self.n_levels,
                                                                  self.n_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            sampling_locations = reference_points[:, :, None, :, None, :] \
                + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] \
                + sampling_offsets / self.n_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index,
                                            sampling_locations, attention_weights, self.im2col_step)
        return self.output_proj(output)
