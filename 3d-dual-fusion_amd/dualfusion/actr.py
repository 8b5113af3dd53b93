"""ACTR: the dual-query deformable cross-attention encoder that fuses camera features into LiDAR
voxel queries (reference: CP/det3d/models/model_utils/actr.py:40-187, build() :619-657;
actr_transformer.py:22-141 (DeformableTransformerACTR), :276-336 (single-query layer), :338-426
(dual-query fusion layer), :428-511 (encoder); attentions.py:34-117 (gates); position_encoding.py).

Module tree and parameter names/shapes equal the reference's so its
checkpoints load:  input_proj.0.{0,1}, i_input_proj.{0,1}, transformer.level_embed,
transformer.encoder.layers.i.{self_attn.*, norm1-3, linear1-4, fusion_layer.{b,a}_conv1d},
transformer.encoder.lidar_attns.i.* (ACTRv2).

What differs is the execution plan (results identical in eval mode):
  * the image-side sine position embedding and padding masks are never materialised: the
    dual-query layer reads neither (value = value_proj(src) has no positional term,
    actr_transformer.py:399-411), masks are all-False so valid_ratios == 1;
  * query-side tensors stay [N, Q, C] row-major; 1x1 Conv1d layers run as GEMMs on that layout
    instead of permute -> conv -> permute;
  * multi-scale deformable sampling is the HIP kernel csrc/msda.hip.
"""
import copy
import math

import os

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import normal_

from .linear_rows import Linear, linear as _linear
from .msda import MSDeformAttn


# ----------------------------------------------------------------------------- gates
class _Gate1D(nn.Module):
    """Two 1x1 Conv1d(C -> 1) gates.  Parameter names b_conv1d / a_conv1d as in attentions.py."""

    def __init__(self, g_channel, g_channel_):
        super().__init__()
        self.g_channel, self.g_channel_ = g_channel, g_channel_
        self.b_conv1d = nn.Conv1d(g_channel, 1, kernel_size=1, stride=1, padding=0)
        self.a_conv1d = nn.Conv1d(g_channel_, 1, kernel_size=1, stride=1, padding=0)

    def _maps(self, x1, x2):
        """sigmoid(conv1x1) on [N, Q, C] inputs without the permutes: one GEMV each."""
        m1 = torch.sigmoid(_linear(x1, self.b_conv1d.weight[:, :, 0], self.b_conv1d.bias))
        m2 = torch.sigmoid(_linear(x2, self.a_conv1d.weight[:, :, 0], self.a_conv1d.bias))
        return m1, m2


class BiGate1D(_Gate1D):
    def forward(self, feat1, feat2):          # attentions.py:34-52
        s1, s2 = self._maps(feat1, feat2)
        return feat1 * s2, feat2 * s1


class BiGate1D_2(_Gate1D):
    def forward(self, feat1, feat2):          # attentions.py:54-74
        fuse = feat1 + feat2
        s1, s2 = self._maps(fuse, fuse)
        return feat1 * s1, feat2 * s2


class BiGateSum1D(_Gate1D):
    def forward(self, feat1, feat2):          # attentions.py:76-94
        s1, s2 = self._maps(feat1, feat2)
        return feat1 + feat2 * s1, feat2 + feat1 * s2


class BiGateSum1D_2(_Gate1D):
    def forward(self, feat1, feat2):          # attentions.py:96-117
        fuse = feat1 + feat2
        s1, s2 = self._maps(fuse, fuse)
        return feat1 + feat2 * s1, feat2 + feat1 * s2


attn_dict = {'BiGate1D': BiGate1D, 'BiGate1D_2': BiGate1D_2, 'BiGateSum1D': BiGateSum1D,
             'BiGateSum1D_2': BiGateSum1D_2}


# ----------------------------------------------------------------------------- position encodings
def _sine_table(x, num_pos_feats, temperature):
    """x [..]: interleaved sin/cos over num_pos_feats channels (position_encoding.py:41-48)."""
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=x.device)
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / num_pos_feats)
    p = x[..., None] / dim_t
    return torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=-1).flatten(-2)


class PositionEmbeddingSine(nn.Module):
    """Image-plane sine embedding (position_encoding.py:17-53); input (tensor, mask)."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask):
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = (y_embed - 0.5) / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = (x_embed - 0.5) / (x_embed[:, :, -1:] + eps) * self.scale
        pos_x = _sine_table(x_embed, self.num_pos_feats, self.temperature)
        pos_y = _sine_table(y_embed, self.num_pos_feats, self.temperature)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class PositionEmbeddingSineSparse(PositionEmbeddingSine):
    """position_encoding.py:56-89: normalised image coordinates of each query -> [N, C, Q]."""

    def forward(self, coor, depth=None):
        x_embed, y_embed = coor[..., 0], coor[..., 1]
        if self.normalize:
            y_embed = y_embed * self.scale
            x_embed = x_embed * self.scale
        pos_x = _sine_table(x_embed, self.num_pos_feats, self.temperature)
        pos_y = _sine_table(y_embed, self.num_pos_feats, self.temperature)
        return torch.cat((pos_y, pos_x), dim=2).permute(0, 2, 1)


class PositionEmbeddingSineSparseDepth(PositionEmbeddingSine):
    """position_encoding.py:91-120: "depth" = LiDAR-frame x of the voxel, / 60 * 2pi -> [N, C, Q]."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__(num_pos_feats, temperature, normalize, scale)
        self.norm_param = 60.

    def forward(self, depth):
        d = depth / self.norm_param * self.scale if self.normalize else depth
        return _sine_table(d, self.num_pos_feats, self.temperature).permute(0, 2, 1)


class PositionEmbeddingLearnedDepth(nn.Module):
    """position_encoding.py:122-141: learned table over num_bin depth bins (depth / 60 * num_bin).
    (The reference's forward takes (feat, depth) although ACTR.forward calls it with one argument,
    actr.py:163; 'depth_learn' is not used by any 3D-DF config.)"""

    def __init__(self, num_pos_feats=256, num_bin=120):
        super().__init__()
        self.d_embed = nn.Embedding(num_bin, num_pos_feats)
        self.num_bin = num_bin
        nn.init.uniform_(self.d_embed.weight)

    def forward(self, depth, feat=None):
        d = (depth / 60. * self.num_bin).to(torch.long)
        return self.d_embed(d).permute(0, 2, 1)


# ----------------------------------------------------------------------------- encoder layers
def _ffn_branch(lin_a, lin_b, x, activation, drop):
    """lin_b(drop(activation(lin_a(x)))): the two linears of a feed-forward block around its [rows x d_ffn] hidden tensor
    (0.5 - 1 GB in a training step).  Under grad:
      * bf16 mixed precision (ops.CONV_PRECISION == "bf16": BASELINE configs[2] / [3]): hidden rows, their activation / dropout
        and both products in bfloat16 (linear_rows.linear_bf16: half the bytes of every pass, 16-bit matrix rate, the weight
        gradients as per-sample batched products), the result back in fp32 for the residual LayerNorm; DF3D_FFN_AUTOCAST=0
        keeps fp32;
      * otherwise with ReLU: activation + dropout as one in-place pass forward and one pass backward without a mask tensor
        (ops.relu_dropout_); the first linear runs on the flattened rows so that its result is a fresh tensor, not a view."""
    from . import ops as _ops
    if torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
        if _ops.CONV_PRECISION == "bf16" and os.environ.get("DF3D_FFN_AUTOCAST", "1") != "0":
            from .linear_rows import linear_bf16
            h = linear_bf16(x.reshape(-1, x.shape[-1]), lin_a.weight, lin_a.bias)          # (flattened: a fresh tensor)
            if activation is F.relu and h.requires_grad and h._base is None and _ops.relu_dropout_supported(h):
                h = _ops.relu_dropout_(h, drop.p if drop.training else 0.0)
            else:
                h = drop(activation(h))
            return linear_bf16(h.view(x.shape[:-1] + (h.shape[-1],)), lin_b.weight, lin_b.bias).float()
        if activation is F.relu and _ops.relu_dropout_supported(x):
            h = lin_a(x.reshape(-1, x.shape[-1]))
            if h.requires_grad and h.is_contiguous() and h._base is None:
                h = _ops.relu_dropout_(h, drop.p if drop.training else 0.0)
            else:
                h = drop(activation(h))
            return lin_b(h.view(x.shape[:-1] + (h.shape[-1],)))
    return lin_b(drop(activation(lin_a(x))))


def _residual_norm(x, y, norm, drop):
    """norm(x + drop(y)): one row kernel each way under grad (ops.dropout_add_layernorm), the modules otherwise."""
    from . import ops as _ops
    return _ops.dropout_add_layernorm(x, y, norm, drop)


def _get_activation_fn(activation):
    if activation == "relu":
        return F.relu
    if activation == "gelu":
        return F.gelu
    if activation == "glu":
        return F.glu
    raise RuntimeError("activation should be relu/gelu, not %s." % activation)


class DeformableTransformerEncoderLayer(nn.Module):
    """Single-query layer (feature_modal 'lidar' / 'image'), actr_transformer.py:276-336."""

    def __init__(self, d_model=256, q_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4, hybrid_cfg=None):
        super().__init__()
        self.d_model = d_model
        self.self_attn = MSDeformAttn(d_model, q_model, n_levels, n_heads, n_points)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, q_pos=None,
                q_feat=None, q_i_feat=None):
        query = q_feat if q_pos is None else q_feat + q_pos
        att = self.self_attn(query, reference_points, src, spatial_shapes, level_start_index, padding_mask)
        q_feat = _residual_norm(q_feat, att, self.norm1, self.dropout1)
        ffn = _ffn_branch(self.linear1, self.linear2, q_feat, self.activation, self.dropout2)
        q_feat = _residual_norm(q_feat, ffn, self.norm2, self.dropout3)
        return q_feat, q_i_feat


class DeformableTransformerFusionEncoderLayer(nn.Module):
    """Dual-query layer, actr_transformer.py:338-426: sampling offsets from the LiDAR query,
    attention weights from LiDAR+image queries, MSDA output added to the IMAGE query stream, one
    FFN per stream, then a bidirectional gate."""

    def __init__(self, d_model=256, q_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8,
                 n_points=4, hybrid_cfg=None):
        super().__init__()
        self.attn_layer = hybrid_cfg['attn_layer']
        self.q_method = hybrid_cfg.get('q_method', None)
        self.q_rep_place = hybrid_cfg.get('q_rep_place', None)
        # Where the bidirectional gate sits differs between the reference's source trees: CenterPoint / TransFusion run
        # the two FFNs first and gate last (CP/det3d/models/model_utils/actr_transformer.py:417-425), Voxel-RCNN gates
        # right after the attention and runs the FFNs on the gated streams (VR/pcdet/models/model_utils/
        # actr_transformer.py:503-512).  Not configurable in the reference (two copies of the file); here a key of
        # hybrid_cfg that `VoxelBackBone8xFusion` sets.  Found by the golden of the reference's own VR fusion glue.
        self.gate_before_ffn = bool(hybrid_cfg.get('gate_before_ffn', False))
        self.d_model = d_model
        self.self_attn = MSDeformAttn(d_model, q_model, n_levels, n_heads, n_points, q_method=self.q_method,
                                      q_rep_place=self.q_rep_place)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = Linear(d_model, d_ffn)      # image-query FFN
        self.activation = _get_activation_fn(activation)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = nn.LayerNorm(d_model)
        self.linear3 = Linear(d_model, d_ffn)      # LiDAR-query FFN
        self.dropout4 = nn.Dropout(dropout)
        self.linear4 = Linear(d_ffn, d_model)
        self.dropout5 = nn.Dropout(dropout)
        self.norm3 = nn.LayerNorm(d_model)
        self.fusion_layer = attn_dict[self.attn_layer](q_model, q_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None, q_pos=None,
                q_feat=None, q_i_feat=None):
        if self._fusable(src, reference_points, padding_mask, q_pos, q_feat, q_i_feat):
            return self._forward_fused(src, reference_points, spatial_shapes, level_start_index, q_pos, q_feat,
                                       q_i_feat)
        lq = q_feat if q_pos is None else q_feat + q_pos
        iq = q_i_feat if q_pos is None else q_i_feat + q_pos
        att = self.self_attn(lq, reference_points, src, spatial_shapes, level_start_index, padding_mask, i_query=iq)
        q_i_feat = _residual_norm(q_i_feat, att, self.norm1, self.dropout1)
        if self.gate_before_ffn:
            q_feat, q_i_feat = self.fusion_layer(q_feat, q_i_feat)
        q_i_feat = _residual_norm(q_i_feat, _ffn_branch(self.linear1, self.linear2, q_i_feat, self.activation, self.dropout2),
                                  self.norm2, self.dropout3)
        q_feat = _residual_norm(q_feat, _ffn_branch(self.linear3, self.linear4, q_feat, self.activation, self.dropout4),
                                self.norm3, self.dropout5)
        if self.gate_before_ffn:
            return q_feat, q_i_feat
        return self.fusion_layer(q_feat, q_i_feat)


    # ---- inference fast path: the ~25 element-wise launches of one layer as 4 HIP kernels (csrc/actr.hip)
    def _fusable(self, src, reference_points, padding_mask, q_pos, q_feat, q_i_feat):
        return (not self.training and not torch.is_grad_enabled() and q_feat.is_cuda and q_pos is not None
                and q_i_feat is not None and padding_mask is None and q_feat.dtype == torch.float32
                and self.q_method == 'sum' and list(self.q_rep_place) == ['weight']
                and self.attn_layer == 'BiGateSum1D_2' and self.activation is F.relu
                and reference_points.shape[-1] == 2 and q_feat.shape[-1] % 4 == 0)

    @staticmethod
    def _linear_relu(lin, x):
        """relu(x W^T + b) with the ReLU in the GEMM epilogue (hipBLASLt) instead of a second pass."""
        return torch._addmm_activation(lin.bias, x.reshape(-1, x.shape[-1]), lin.weight.t()).view(
            x.shape[:-1] + (lin.out_features,))

    def fusable_config(self):
        return (not self.training and self.q_method == 'sum' and list(self.q_rep_place) == ['weight']
                and self.attn_layer == 'BiGateSum1D_2' and self.activation is F.relu and self.d_model % 4 == 0)

    def _ffn(self, x, lin_a, lin_b, norm, slot):
        """norm(x + lin_b(relu(lin_a(x)))): one fused kernel (csrc/ffn.hip) when the sizes fit it."""
        from . import ops as _ops
        if not _ops.ffn_supported(lin_a.in_features, lin_a.out_features):
            return _ops.add_layernorm(x, lin_b(self._linear_relu(lin_a, x)), norm.weight, norm.bias, norm.eps)
        key = (lin_a.weight.data_ptr(), lin_a.weight._version, lin_b.weight.data_ptr(), lin_b.weight._version)
        hit = getattr(self, slot, None)
        if hit is None or hit[0] != key:
            hit = (key, _ops.ffn_pack(lin_a.weight.detach().contiguous(), lin_b.weight.detach().contiguous()))
            object.__setattr__(self, slot, hit)
        return _ops.ffn_fused(x, hit[1], lin_a.bias, lin_b.bias, lin_a.out_features, residual=x,
                              ln_weight=norm.weight, ln_bias=norm.bias, eps=norm.eps)

    def _ffn_pair(self, qi, q):
        """The image-query FFN (linear1/2, norm2) and the LiDAR-query FFN (linear3/4, norm3) as ONE launch."""
        from . import ops as _ops
        if not _ops.ffn_supported(self.linear1.in_features, self.linear1.out_features) or \
                self.linear1.out_features != self.linear3.out_features:
            return (self._ffn(qi, self.linear1, self.linear2, self.norm2, "_ffn_i"),
                    self._ffn(q, self.linear3, self.linear4, self.norm3, "_ffn_p"))
        jobs = []
        for x, la, lb, norm, slot in ((qi, self.linear1, self.linear2, self.norm2, "_ffn_i"),
                                      (q, self.linear3, self.linear4, self.norm3, "_ffn_p")):
            key = (la.weight.data_ptr(), la.weight._version, lb.weight.data_ptr(), lb.weight._version)
            hit = getattr(self, slot, None)
            if hit is None or hit[0] != key:
                hit = (key, _ops.ffn_pack(la.weight.detach().contiguous(), lb.weight.detach().contiguous()))
                object.__setattr__(self, slot, hit)
            jobs.append(dict(x=x.contiguous(), packed=hit[1], b1=la.bias, b2=lb.bias, residual=x.contiguous(),
                             ln_weight=norm.weight, ln_bias=norm.bias, eps=norm.eps))
        return tuple(_ops.ffn_fused_jobs(jobs, self.linear1.out_features))

    def _forward_fused(self, src, reference_points, spatial_shapes, level_start_index, q_pos, q_feat, q_i_feat,
                       value=None):
        """Same arithmetic as forward(); `reference_points` [N,Q,L,2] carries the same (x, y) for every level when
        the valid ratios are 1 (ACTR feeds unpadded maps), which is what the fused sampler assumes."""
        from . import ops as _ops
        sa = self.self_attn
        q_feat, q_i_feat, q_pos = q_feat.contiguous(), q_i_feat.contiguous(), q_pos.contiguous()
        pixel_scale = image_bias = None
        if value is None:
            value = sa.project_value(src)
        elif isinstance(value, tuple):
            value, pixel_scale, image_bias = value
        ref_xy = reference_points[:, :, 0, :].contiguous()
        C = q_feat.shape[-1]
        n_off, n_w = sa.sampling_offsets.out_features, sa.attention_weights.out_features
        if (_ops.CONV_PRECISION != "fp32" and n_off % 16 == 0 and _ops.rows_linear_supported(C, n_off + n_w)
                and _ops.rows_linear_supported(C, C) and sa.output_proj.out_features == C and C % 16 == 0):
            # the two query mixtures are formed on load and both linears run as one launch on the matrix cores; the
            # output projection carries its residual + LayerNorm (csrc/rowlinear.hip): 2 launches instead of 5
            pk = _ops.rows_linear_pack([sa.sampling_offsets.weight, sa.attention_weights.weight],
                                       [sa.sampling_offsets.bias, sa.attention_weights.bias])
            offsets, logits = _ops.rows_linear(q_feat, pk, x1=q_i_feat, x2=q_pos, csplit=n_off, n0=n_off, n1=n_w)
            out = _ops.ms_deform_attn_fused(value, spatial_shapes, level_start_index, ref_xy, offsets, logits, sa.n_levels,
                                            sa.n_points, pixel_scale, image_bias)
            qi = _ops.rows_linear(out.contiguous(), _ops.rows_linear_pack(sa.output_proj.weight, sa.output_proj.bias),
                                  ln=(q_i_feat, self.norm1))
        else:
            A, Bw = _ops.actr_prep(q_feat, q_i_feat, q_pos)
            out = _ops.ms_deform_attn_fused(value, spatial_shapes, level_start_index, ref_xy, sa.sampling_offsets(A),
                                            sa.attention_weights(Bw), sa.n_levels, sa.n_points, pixel_scale, image_bias)
            att = sa.output_proj(out)
            qi = _ops.add_layernorm(q_i_feat, att, self.norm1.weight, self.norm1.bias, self.norm1.eps)
        g = self.fusion_layer
        if self.gate_before_ffn:
            q, qi = _ops.bigate_sum(q_feat, qi, g.b_conv1d.weight.view(-1), g.b_conv1d.bias, g.a_conv1d.weight.view(-1),
                                    g.a_conv1d.bias)
            qi, q = self._ffn_pair(qi, q)
            return q, qi
        qi, q = self._ffn_pair(qi, q_feat)
        return _ops.bigate_sum(q, qi, g.b_conv1d.weight.view(-1), g.b_conv1d.bias, g.a_conv1d.weight.view(-1),
                               g.a_conv1d.bias)


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class DeformableTransformerEncoder(nn.Module):
    """actr_transformer.py:428-511.  ACTRv2 runs a LocalTransformer (3-D local self-attention over
    the LiDAR queries) before every deformable layer (:496-498)."""

    def __init__(self, encoder_layer, num_layers, model_name='ACTR', lt_cfg=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.model_name = model_name
        if model_name == 'ACTRv2':
            from .pointformer import LocalTransformer
            self.lidar_attns = _get_clones(
                LocalTransformer(lt_cfg['npoint'], lt_cfg['radius'], lt_cfg['nsample'], encoder_layer.d_model,
                                 encoder_layer.d_model, num_layers=lt_cfg['num_layers'],
                                 attn_feat_agg_method=lt_cfg.get('attn_feat_agg_method', 'unique'),
                                 feat_agg_method=lt_cfg.get('feat_agg_method', 'replace')), num_layers)

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None, q_feat=None,
                q_pos=None, q_reference_points=None, q_lidar_grid=None, q_i_feat=None, layer_values=None):
        """`layer_values` (inference fast path): per layer (value [N,S,M,D], pixel_scale, image_bias) already
        projected by ACTR.forward_folded; `src` is then unused and valid_ratios are 1."""
        if q_reference_points is None:
            raise NotImplementedError("IACTR (image-grid queries) is not part of the 3D-DF configs")
        if layer_values is not None:
            reference_points = q_reference_points[:, :, None]
        else:
            reference_points = q_reference_points[:, :, None] * valid_ratios[:, None]
        for idx, layer in enumerate(self.layers):
            if self.model_name == 'ACTRv2':
                q_feat = self.lidar_attns[idx](q_lidar_grid, q_feat.permute(0, 2, 1))
            if layer_values is not None:
                q_feat, q_i_feat = layer._forward_fused(None, reference_points, spatial_shapes, level_start_index,
                                                        q_pos, q_feat, q_i_feat, value=layer_values[idx])
            else:
                q_feat, q_i_feat = layer(src, pos, reference_points, spatial_shapes, level_start_index, padding_mask,
                                         q_pos=q_pos, q_feat=q_feat, q_i_feat=q_i_feat)
        return q_feat


class DeformableTransformerACTR(nn.Module):
    def __init__(self, d_model=256, query_num_feat=256, nhead=8, num_encoder_layers=6, dim_feedforward=1024,
                 dropout=0.1, activation="relu", return_intermediate_dec=False, num_feature_levels=4, enc_n_points=4,
                 two_stage=False, two_stage_num_proposals=300, model_name='ACTR', lt_cfg=None, feature_modal='lidar',
                 hybrid_cfg=None):
        super().__init__()
        self.d_model = d_model
        self.q_model = query_num_feat
        self.nhead = nhead
        self.two_stage = two_stage
        self.two_stage_num_proposals = two_stage_num_proposals
        self.feature_modal = feature_modal
        layer_cls = DeformableTransformerFusionEncoderLayer if feature_modal in ['hybrid'] \
            else DeformableTransformerEncoderLayer
        encoder_layer = layer_cls(self.d_model, self.q_model, dim_feedforward, dropout, activation,
                                  num_feature_levels, nhead, enc_n_points, hybrid_cfg)
        self.encoder = DeformableTransformerEncoder(encoder_layer, num_encoder_layers, model_name=model_name,
                                                    lt_cfg=lt_cfg)
        self.level_embed = nn.Parameter(torch.Tensor(num_feature_levels, d_model))
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        normal_(self.level_embed)

    def forward(self, srcs, masks, pos_embeds, q_feat_flatten, q_pos, q_ref_coors, q_lidar_grid=None,
                q_i_feat_flatten=None):
        """srcs: list of [N, C, H, W].  masks / pos_embeds may be None (all-valid maps): the ACTR
        layers consume neither the masks' padding (all False) nor the level position embedding."""
        dev = srcs[0].device
        shapes = [(int(s.shape[2]), int(s.shape[3])) for s in srcs]
        # contiguous [N, sum HW, C]: value_proj on a transposed view makes hipBLASLt pick a 25x slower kernel
        src_flatten = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1).contiguous()
        # cached by value: a fresh host -> device copy of this tiny tensor is a pageable-memory copy, i.e. the host waits for
        # everything queued on the stream before it (1.4 ms per step in the Voxel-RCNN tree)
        skey = (tuple(shapes), str(dev))
        hit = self.__dict__.get("_level_cache")
        if hit is None or hit[0] != skey:
            spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=dev)
            hit = (skey, spatial_shapes,
                   torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1])))
            self.__dict__["_level_cache"] = hit
        spatial_shapes, level_start_index = hit[1], hit[2]
        N = src_flatten.shape[0]
        if masks is None or all(m is None for m in masks):
            valid_ratios = torch.ones((N, len(srcs), 2), dtype=torch.float32, device=dev)
            mask_flatten = None
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
            mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
            if not bool(mask_flatten.any()):
                mask_flatten = None
        lvl_pos = None
        if pos_embeds is not None:
            lvl_pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[l].view(1, 1, -1)
                                 for l, p in enumerate(pos_embeds)], 1)
        return self.encoder(src_flatten, spatial_shapes, level_start_index, valid_ratios, lvl_pos, mask_flatten,
                            q_pos=q_pos, q_feat=q_feat_flatten, q_reference_points=q_ref_coors,
                            q_lidar_grid=q_lidar_grid, q_i_feat=q_i_feat_flatten)

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)


class ACTR(nn.Module):
    """actr.py:40-187.  forward(v_feat [N,Q,C], grid [N,Q,2] in [0,1], i_feats [[N,Cimg,H,W]..],
    v_i_feat [N,Q,Cimg], lidar_grid [N,Q,3]) -> enhanced LiDAR queries [N,Q,C]."""

    def __init__(self, transformer, num_channels, num_feature_levels, max_num_ne_voxel, p_num_channels=None,
                 pos_encode_method="image_coor", feature_modal='lidar'):
        super().__init__()
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.num_feature_levels = num_feature_levels
        self.num_backbone_outs = len(num_channels)
        assert self.num_backbone_outs == num_feature_levels
        self.input_proj = nn.ModuleList([
            nn.Sequential(nn.Conv2d(num_channels[l], hidden_dim, kernel_size=1), nn.GroupNorm(32, hidden_dim))
            for l in range(max(num_feature_levels, 1))])
        for proj in self.input_proj:
            nn.init.xavier_uniform_(proj[0].weight, gain=1)
            nn.init.constant_(proj[0].bias, 0)
        if feature_modal in ['image', 'hybrid']:
            self.i_input_proj = nn.Sequential(nn.Conv1d(num_channels[0], hidden_dim, kernel_size=1),
                                              nn.GroupNorm(32, hidden_dim))
            nn.init.xavier_uniform_(self.i_input_proj[0].weight, gain=1)
            nn.init.constant_(self.i_input_proj[0].bias, 0)
        self.feature_modal = feature_modal
        self.max_num_ne_voxel = max_num_ne_voxel
        self.pos_encode_method = pos_encode_method
        assert self.pos_encode_method in ["image_coor", "depth", "depth_learn"]
        if pos_encode_method == "image_coor":
            self.q_position_embedding = PositionEmbeddingSineSparse(num_pos_feats=transformer.q_model // 2,
                                                                    normalize=True)
        elif pos_encode_method == "depth":
            self.q_position_embedding = PositionEmbeddingSineSparseDepth(num_pos_feats=transformer.q_model,
                                                                         normalize=True)
        else:
            self.q_position_embedding = PositionEmbeddingLearnedDepth(num_pos_feats=transformer.q_model)
        self.v_position_embedding = PositionEmbeddingSine(num_pos_feats=hidden_dim // 2, normalize=True)

    def _input_proj(self, l, src):
        """input_proj[l] = Conv2d 1x1 + GroupNorm.  The 1x1 conv runs as one batched GEMM: MIOpen without a tuning
        database falls back to its naive fp32 kernel for this shape (6.7 ms per call at KITTI size)."""
        conv, gn = self.input_proj[l][0], self.input_proj[l][1]
        if src.is_cuda and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1) and conv.groups == 1:
            N, C, H, W = src.shape
            if torch.is_grad_enabled():
                # differentiable; the weight gradient as chunked batched products, no transposed clone of the maps
                # (ops._ChannelFirstLinear)
                from . import ops as _ops
                y = _ops.channel_first_linear(src.reshape(N, C, H * W), conv.weight[:, :, 0, 0])
            else:
                # (inference: bmm, not torch.matmul -- matmul of a 2-D weight with a 3-D map clones the map transposed and
                # transposes the result back: 0.4 ms per step at the Voxel-RCNN size)
                y = torch.bmm(conv.weight[:, :, 0, 0].unsqueeze(0).expand(N, -1, -1), src.reshape(N, C, H * W))
            if conv.bias is not None:
                y = y + conv.bias[None, :, None]
            return gn(y.view(N, -1, H, W))
        return self.input_proj[l](src)

    def project_image_queries(self, v_i_feat):
        """i_input_proj on [N, Q, Cimg]: the 1x1 Conv1d as a GEMM, GroupNorm over (group, Q)."""
        conv, gn = self.i_input_proj[0], self.i_input_proj[1]
        from . import ops as _ops
        if (v_i_feat.is_cuda and v_i_feat.dtype == torch.float32 and not torch.is_grad_enabled() and v_i_feat.shape[1] > 0
                and _ops.CONV_PRECISION != "fp32" and _ops.rows_linear_supported(conv.in_channels, conv.out_channels)):
            y = _ops.rows_linear(v_i_feat.contiguous(), _ops.rows_linear_pack(conv.weight, conv.bias))
        else:
            y = _linear(v_i_feat, conv.weight[:, :, 0], conv.bias)        # [N, Q, C]
        if y.is_cuda and y.dtype == torch.float32 and not torch.is_grad_enabled() \
                and (gn.num_channels // gn.num_groups) % 4 == 0 and y.shape[1] > 0:
            from . import ops as _ops
            return _ops.rows_groupnorm(y.contiguous(), gn)
        y = F.group_norm(y.transpose(1, 2), gn.num_groups, gn.weight, gn.bias, gn.eps)
        return y.transpose(1, 2)

    def forward_projected(self, v_feat, grid, src_conv, v_i_feat, lidar_grid, q_pos=None):
        """forward() for callers that already hold the 1x1 input projection of the (single-level) image,
        `src_conv` [N, C, H, W] = input_proj[0][0](image) before its GroupNorm, and optionally the query
        position embedding [N, Q, C].  Used by the CenterPoint adapter, which folds its image gate into
        that projection (dualfusion/fusion.py)."""
        q_feat = v_feat
        q_i_feat = None
        if self.feature_modal in ['image', 'hybrid']:
            q_i_feat = self.project_image_queries(v_i_feat)
            if self.feature_modal == 'image':
                q_feat = q_i_feat
        if q_pos is None:
            if self.pos_encode_method == "image_coor":
                q_pos = self.q_position_embedding(grid).transpose(1, 2)
            else:
                q_pos = self.q_position_embedding(lidar_grid[..., 0]).transpose(1, 2)
        srcs = [self.input_proj[0][1](src_conv)]
        return self.transformer(srcs, None, None, q_feat, q_pos, grid, q_lidar_grid=lidar_grid,
                                q_i_feat_flatten=q_i_feat)

    def can_fold(self):
        """True when forward_folded() applies: inference, one image level, dual-query layers of the 3D-DF config."""
        layers = self.transformer.encoder.layers
        return (not self.training and not torch.is_grad_enabled() and self.feature_modal == 'hybrid'
                and self.num_feature_levels == 1 and self.input_proj[0][1].num_channels <= 256
                and all(isinstance(l, DeformableTransformerFusionEncoderLayer) and l.fusable_config()
                        for l in layers))

    def _value_weights(self):
        layers = self.transformer.encoder.layers
        vkey = tuple((l.self_attn.value_proj.weight.data_ptr(), l.self_attn.value_proj.weight._version,
                      l.self_attn.value_proj.bias._version) for l in layers)
        hit = getattr(self, "_vcat", None)
        if hit is None or hit[0] != vkey:
            hit = (vkey, torch.cat([l.self_attn.value_proj.weight for l in layers], 0).contiguous(),
                   torch.cat([l.self_attn.value_proj.bias for l in layers], 0).contiguous())
            object.__setattr__(self, "_vcat", hit)
        return hit[1], hit[2]

    def start_values(self, u, gate):
        """The image side of forward_folded (moments of the gated projection, GroupNorm fold, value rows of every layer) queued
        on a stream of its own as soon as the gate exists: it streams 370 MB and depends on nothing the query side produces
        (query assembly, image-query projection + GroupNorm, the first offset / weight linear: ~150 us of gathers and small
        launches that leave the memory system idle), so the two run beside each other and meet at the first sampler.
        Returns what forward_folded takes as `values`.  Outputs come from the CURRENT stream's pool (allocated before the
        fork event, last used and freed behind the join event): no allocator traffic on the side stream."""
        from . import ops as _ops
        conv, gn = self.input_proj[0][0], self.input_proj[0][1]
        W, wb = self._value_weights()
        dev = u.device
        side = self.__dict__.get("_value_stream")
        if side is None:
            side = torch.cuda.Stream(device=dev)
            object.__setattr__(self, "_value_stream", side)
        main = torch.cuda.current_stream(dev)
        fork = torch.cuda.Event()
        # the allocations happen inside value_fold_gemm on the current stream's pool, BEFORE its launches: a recycled block's
        # earlier users were queued on `main` ahead of this event
        value_cf = [None]
        fork.record(main)
        side.wait_event(fork)
        value_cf[0] = _ops.value_fold_gemm(u, gate, conv.bias, gn, W, wb, bf16=_ops.CONV_PRECISION == "bf16", stream=side)
        join = torch.cuda.Event()
        join.record(side)
        return value_cf[0][0], value_cf[0][1], join

    def forward_folded(self, v_feat, grid, u, gate, hw, v_i_feat, lidar_grid, q_pos=None, values=None):
        """Inference path that never materialises the normalised image map.  `u` [N, >=C, H*W] channel-first is
        input_proj[0][0] WITHOUT its bias applied to the (un-gated) image, `gate` [N, H*W] the adapter's per-pixel
        image gate (or None).  input_proj's GroupNorm and every layer's value_proj are folded into one per-image
        GEMM (csrc/actr.hip gn_fold_kernel); the gate and the folded constant are applied inside the sampler.
        values: the result of start_values(u, gate) (the image side already queued on its own stream)."""
        from . import ops as _ops
        conv, gn = self.input_proj[0][0], self.input_proj[0][1]
        layers = self.transformer.encoder.layers
        C = gn.num_channels
        W, wb = self._value_weights()
        join = None
        if values is not None:
            value, cf, join = values
        elif u.dtype == torch.uint8:
            # u arrives as pixel-major split rows (csrc/imgproj.hip): moments, fold and the value GEMM of all
            # layers in one native call on the bf16 matrix cores
            # reduced-precision mode (DF3D_CONV_PRECISION=bf16, the reference's fp16-AMP configurations): the value rows
            # the sampler gathers are bf16 -- half the bytes; weights, accumulation and output stay fp32
            value, cf = _ops.value_fold_gemm(u, gate, conv.bias, gn, W, wb, bf16=_ops.CONV_PRECISION == "bf16")
        else:
            Wf, cf = _ops.groupnorm_fold(u, gate, conv.bias, gn, W, wb)
            value = torch.bmm(u[:, :C].transpose(1, 2), Wf.transpose(1, 2))         # [N, S, nlayers*C]
        M = layers[0].self_attn.n_heads
        layer_values = [(value[:, :, i * C:(i + 1) * C].unflatten(-1, (M, C // M)), gate, cf[:, i * C:(i + 1) * C])
                        for i in range(len(layers))]
        q_i_feat = self.project_image_queries(v_i_feat)
        if q_pos is None:
            if self.pos_encode_method == "image_coor":
                q_pos = self.q_position_embedding(grid).transpose(1, 2)
            else:
                q_pos = self.q_position_embedding(lidar_grid[..., 0]).transpose(1, 2)
        skey = (int(hw[0]), int(hw[1]), str(u.device))
        shp = getattr(self, "_shape_cache", None)
        if shp is None or shp[0] != skey:
            spatial_shapes = torch.as_tensor([hw], dtype=torch.long, device=u.device)
            shp = (skey, spatial_shapes, spatial_shapes.new_zeros((1,)))
            object.__setattr__(self, "_shape_cache", shp)
        spatial_shapes, level_start_index = shp[1], shp[2]
        if join is not None:
            torch.cuda.current_stream(u.device).wait_event(join)       # the value rows: first read by the first sampler
        return self.transformer.encoder(None, spatial_shapes, level_start_index, None, q_feat=v_feat, q_pos=q_pos,
                                        q_reference_points=grid, q_lidar_grid=lidar_grid, q_i_feat=q_i_feat,
                                        layer_values=layer_values)

    def forward(self, v_feat, grid, i_feats, v_i_feat=None, lidar_grid=None):
        q_feat = v_feat
        q_i_feat = None
        if self.feature_modal in ['image', 'hybrid']:
            assert v_i_feat is not None
            q_i_feat = self.project_image_queries(v_i_feat)
            if self.feature_modal == 'image':
                q_feat = q_i_feat
        if self.pos_encode_method == "image_coor":
            q_pos = self.q_position_embedding(grid).transpose(1, 2)
        else:
            q_pos = self.q_position_embedding(lidar_grid[..., 0]).transpose(1, 2)
        srcs = [self._input_proj(l, src) for l, src in enumerate(i_feats)]
        return self.transformer(srcs, None, None, q_feat, q_pos, grid, q_lidar_grid=lidar_grid,
                                q_i_feat_flatten=q_i_feat)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def build(model_cfg, model_name='ACTR', lt_cfg=None, hybrid_cfg=None):
    """actr.py:619-657 (the Voxel-RCNN tree passes hybrid_cfg separately, VR/.../actr.py:619) with the argparse defaults it relies on (actr_utils.py:548-646): nheads 8,
    enc_n_points 4, dim_feedforward 1024, dropout 0.1."""
    if model_name not in ('ACTR', 'ACTRv2'):
        raise NotImplementedError("%s: only ACTR / ACTRv2 are used by the 3D-DF configs" % model_name)
    get = model_cfg.get if hasattr(model_cfg, "get") else (lambda k, d=None: getattr(model_cfg, k, d))
    num_channels = model_cfg['num_channels'] if isinstance(model_cfg, dict) else model_cfg.num_channels
    q = get('query_num_feat')
    transformer = DeformableTransformerACTR(
        d_model=q, query_num_feat=q, nhead=8, num_encoder_layers=get('num_enc_layers'), dim_feedforward=1024,
        dropout=0.1, activation="relu", return_intermediate_dec=True, num_feature_levels=len(num_channels),
        enc_n_points=4, two_stage=False, two_stage_num_proposals=300, model_name=model_name,
        lt_cfg=_Cfg(lt_cfg) if lt_cfg is not None else None, feature_modal=get('feature_modal', 'lidar'),
        hybrid_cfg=hybrid_cfg if hybrid_cfg is not None else get('hybrid_cfg', None))
    return ACTR(transformer, num_feature_levels=len(num_channels), p_num_channels=get('p_num_channels', None),
                num_channels=num_channels, max_num_ne_voxel=get('max_num_ne_voxel'),
                pos_encode_method=get('pos_encode_method'), feature_modal=get('feature_modal', 'lidar'))
