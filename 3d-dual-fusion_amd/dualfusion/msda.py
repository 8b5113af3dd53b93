"""MSDeformAttn module + autograd Function shell with the reference's names and signatures
(CP/det3d/models/model_utils/ops/modules/ms_deform_attn.py:33-190,
 .../ops/functions/ms_deform_attn_func.py:21-38).  The sampling kernel is csrc/msda.hip."""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.nn.init import xavier_uniform_

from .linear_rows import Linear
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
        self.sampling_offsets = Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = Linear(d_model, d_model)
        self.output_proj = Linear(d_model, d_model)
        self.q_method = q_method
        self.q_rep_place = q_rep_place
        if q_method == 'gating':
            from .actr import attn_dict
            self.q_gating = attn_dict['BiGateSum1D_2'](d_model, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Initial state of ms_deform_attn.py:76-90: zero offset / weight matrices; the offset bias of head m points
        along direction 2*pi*m/n_heads (scaled so its larger component is 1) and grows linearly with the point index;
        Xavier projections with zero bias."""
        H, L, P = self.n_heads, self.n_levels, self.n_points
        with torch.no_grad():
            ang = torch.arange(H, dtype=torch.float32) * (2.0 * math.pi / H)
            ray = torch.stack((ang.cos(), ang.sin()), dim=-1)                      # [H, 2]
            ray = ray / ray.abs().amax(dim=-1, keepdim=True)
            reach = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)   # point p sits p+1 steps out
            bias = ray.view(H, 1, 1, 2).repeat(1, L, P, 1) * reach
            self.sampling_offsets.bias = nn.Parameter(bias.reshape(-1))
            for lin in (self.sampling_offsets, self.attention_weights):
                lin.weight.zero_()
            self.attention_weights.bias.zero_()
            for lin in (self.value_proj, self.output_proj):
                xavier_uniform_(lin.weight)
                lin.bias.zero_()

    def project_value(self, input_flatten, input_padding_mask=None):
        N, Len_in, _ = input_flatten.shape
        value = _ops.linear_rows(input_flatten, self.value_proj)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], float(0))
        return value.view(N, Len_in, self.n_heads, self.d_model // self.n_heads)

    def _queries(self, query, i_query):
        """(query driving the offsets, query driving the weights) after the dual-query mixing rule `q_method`
        (ms_deform_attn.py:129-147): the LiDAR query is replaced by the mixed one in the places `q_rep_place` names."""
        if self.q_method is None:
            return query, query
        assert i_query is not None and self.q_rep_place is not None
        if self.q_method == 'sum':
            mixed = query + i_query
        elif self.q_method == 'image':
            mixed = i_query
        elif self.q_method == 'gating':
            gq, gi = self.q_gating(query, i_query)
            mixed = gq + gi - query - i_query
        else:
            raise NotImplementedError('q_method must be among ["gating", "sum", "image"]')
        return (mixed if 'offset' in self.q_rep_place else query), (mixed if 'weight' in self.q_rep_place else query)

    def _locations(self, reference_points, offsets, spatial_shapes):
        """Normalised sampling locations [N, Lq, M, L, P, 2]: 2-d reference points move by offset / (W_l, H_l);
        4-d ones (cx, cy, w, h) by offset / P * (w, h) / 2."""
        last = reference_points.shape[-1]
        ref = reference_points[:, :, None, :, None, :]
        if last == 2:
            wh = spatial_shapes.flip(-1)                                            # (H, W) -> (W, H)
            return ref + offsets / wh[None, None, None, :, None, :]
        if last == 4:
            return ref[..., :2] + offsets / self.n_points * ref[..., 2:] * 0.5
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(last))

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, i_query=None):
        N, Len_q, _ = query.shape
        H, L, P = self.n_heads, self.n_levels, self.n_points
        assert input_spatial_shapes.shape[0] == L
        value = self.project_value(input_flatten, input_padding_mask)
        q_off, q_w = self._queries(query, i_query)
        offsets = self.sampling_offsets(q_off).view(N, Len_q, H, L, P, 2)
        weights = F.softmax(self.attention_weights(q_w).view(N, Len_q, H, L * P), -1).view(N, Len_q, H, L, P)
        locations = self._locations(reference_points, offsets, input_spatial_shapes)
        output = MSDeformAttnFunction.apply(value, input_spatial_shapes, input_level_start_index, locations, weights,
                                            self.im2col_step)
        return _ops.linear_rows(output, self.output_proj)
