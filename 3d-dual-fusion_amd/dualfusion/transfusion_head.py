"""Detection head of the TransFusion tree: `TransFusionHead`, LiDAR-only branch
(`fuse_img=False`, as in TF/configs/transfusion_nusc_voxel_F.py:244-300), with the reference's constructor arguments,
parameter names (so its checkpoints load) and output structure (TF/mmdet3d/models/dense_heads/transfusion_head.py:
594-1045, 1285-1376; TransFusionBBoxCoder: core/bbox/coders/transfusion_bbox_coder.py).

Device path in eval mode:
  * shared 3x3 conv (512 -> 128) and the two heat-map convs (128 -> 128 + BN + ReLU, 128 -> classes) = three launches
    of the split-precision row kernel on the neck's pixel rows (no NCHW round trip);
  * sigmoid + local-maximum suppression + top-k + query feature / position gather = `df3d_heatmap_proposals`
    (keys -> radix sort -> gather) instead of ~25 launches and four map-sized temporaries;
  * decoder layer: key / value projection of the 32 400 BEV pixels as ONE GEMM on the pixel rows (the learned position
    embedding of the fixed BEV grid is input-independent in eval mode and folded into a cached additive term), the
    200 x 32 400 cross-attention as `df3d_cross_attention` (keys split over workgroups, log-sum-exp merge), all
    prediction heads as two GEMMs (BN folded, block-diagonal second layer);
  * `get_bboxes` = `df3d_transfusion_decode` (one launch).
Training-mode forward runs the plain torch modules (autograd).

`loss` / `get_targets` (transfusion_head.py:1048-1283; round 3): Hungarian target assignment + focal / L1 / Gaussian-focal
losses.  Plain torch formulation (`loss`, autograd) from dualfusion/tf_losses.py; on the device the matching costs
(incl. the rotated 3-D IoU), the Gaussian heat-map targets and the three losses with their gradients are csrc/tfloss.hip
kernels (`loss_device`), the assignment itself `scipy.optimize.linear_sum_assignment` on the host as in the reference."""
import copy
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import ops as _ops
from . import tf_losses as _tl
from .registry import HEADS, MM_HEADS


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule as the head uses it: conv [+ norm] + ReLU, attributes `conv` / `bn` / `activate`,
    bias='auto' = no conv bias when a norm layer follows."""

    def __init__(self, cin, cout, kernel_size, padding=0, bias='auto', conv_cfg=None, norm_cfg=None):
        super(_ConvModule, self).__init__()
        conv = {'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d}[(conv_cfg or dict(type='Conv2d'))['type']]
        self.conv = conv(cin, cout, kernel_size, stride=1, padding=padding,
                         bias=(norm_cfg is None) if bias == 'auto' else bool(bias))
        if norm_cfg is not None:
            self.bn = {'BN1d': nn.BatchNorm1d, 'BN2d': nn.BatchNorm2d, 'BN': nn.BatchNorm2d}[norm_cfg['type']](cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if hasattr(self, "bn"):
            x = self.bn(x)
        return self.activate(x)


class PositionEmbeddingLearned(nn.Module):
    """transfusion_head.py:25-41."""

    def __init__(self, input_channel, num_pos_feats=288):
        super(PositionEmbeddingLearned, self).__init__()
        self.position_embedding_head = nn.Sequential(nn.Conv1d(input_channel, num_pos_feats, kernel_size=1),
                                                     nn.BatchNorm1d(num_pos_feats), nn.ReLU(inplace=True),
                                                     nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class MultiheadAttention(nn.Module):
    """transfusion_head.py:125-253 for kdim = vdim = embed_dim, no key bias / zero attention / masks: parameters
    `in_proj_weight`, `in_proj_bias`, `out_proj`.  Operands are batch-first [N, L, E] here."""

    def __init__(self, embed_dim, num_heads, dropout=0., bias=True):
        super(MultiheadAttention, self).__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.dropout, self.head_dim = embed_dim, num_heads, dropout, embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=bias)
        nn.init.xavier_uniform_(self.in_proj_weight)
        if bias:
            nn.init.constant_(self.in_proj_bias, 0.)
            nn.init.constant_(self.out_proj.bias, 0.)

    def _heads(self, t):
        return t.view(t.shape[0], t.shape[1], self.num_heads, self.head_dim).transpose(1, 2)

    def attend(self, q, k, v):
        """q [N, L, E], k / v [N, S, E] already projected -> out_proj(softmax(q k^T / sqrt(d)) v)."""
        if (torch.is_grad_enabled() and self.head_dim == 16 and q.shape[1] * 16 <= k.shape[1]
                and _ops.cross_attention_train_supported(q, k, v, self.num_heads)):
            # training, a few hundred queries against a BEV map of keys: keys split over waves, probabilities recomputed in the
            # backward (csrc/xattn.hip) -- the [N, heads, L, S] tensors of the composition below are never written
            o = _ops.cross_attention_train(q, k, v, self.num_heads, self.head_dim ** -0.5,
                                          self.dropout if self.training else 0.0)
            return self.out_proj(o)
        qh, kh, vh = self._heads(q), self._heads(k), self._heads(v)
        if (torch.is_grad_enabled() and q.is_cuda and q.shape[1] * 16 <= k.shape[1]
                and os.environ.get("DF3D_TFHEAD_MATH_ATTN", "1") == "1"):
            # training, a few hundred queries against a BEV map of keys: the library's fused attention parallelises its
            # backward over QUERY blocks (200 queries = a handful of workgroups: 2.4 ms for d(query) + 0.8 ms forward per call at
            # 200 x 32 400 x 8 heads x 4 samples); the explicit softmax(q k^T / sqrt(d)) v is batched GEMMs + one softmax over
            # [N, heads, L, S] and differentiates through the same
            p = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * (self.head_dim ** -0.5), -1)
            if self.training and self.dropout > 0:
                p = F.dropout(p, self.dropout)
            o = torch.matmul(p, vh)
        else:
            o = F.scaled_dot_product_attention(qh, kh, vh, dropout_p=self.dropout if self.training else 0.0)
        return self.out_proj(o.transpose(1, 2).reshape(q.shape[0], q.shape[1], self.embed_dim))

    def forward(self, query, key, value):
        E = self.embed_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        bs = (None, None, None) if b is None else (b[:E], b[E:2 * E], b[2 * E:])
        # (the key / value projections see the whole BEV map: under grad their weight gradients contract over B * H * W rows,
        # which the row kernel of linear_rows takes; otherwise this is F.linear)
        from .linear_rows import linear as _lin
        return self.attend(F.linear(query, w[:E], bs[0]), _lin(key, w[E:2 * E], bs[1]), _lin(value, w[2 * E:], bs[2]))


class TransformerDecoderLayer(nn.Module):
    """transfusion_head.py:44-122 (post-norm decoder layer with learned position embeddings on query and key)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", self_posembed=None,
                 cross_posembed=None, cross_only=False):
        super(TransformerDecoderLayer, self).__init__()
        self.cross_only = cross_only
        if not cross_only:
            self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2, self.dropout3 = nn.Dropout(dropout), nn.Dropout(dropout), nn.Dropout(dropout)
        if activation not in ("relu", "gelu"):
            raise RuntimeError("activation should be relu/gelu, not %s." % activation)
        self.activation = F.relu if activation == "relu" else F.gelu
        self.self_posembed, self.cross_posembed = self_posembed, cross_posembed

    def forward(self, query, key, query_pos, key_pos):
        """query [B, C, Pq], key [B, C, Pk], query_pos [B, Pq, 2], key_pos [B, Pk, 2] -> [B, C, Pq]."""
        qpe = self.self_posembed(query_pos).transpose(1, 2) if self.self_posembed is not None else 0
        kpe = self.cross_posembed(key_pos).transpose(1, 2) if self.cross_posembed is not None else 0
        q = query.transpose(1, 2)
        k = key.transpose(1, 2) + kpe
        return self.finish(q, qpe, lambda qq: self.multihead_attn(qq, k, k)).transpose(1, 2)

    def finish(self, q, qpe, cross):
        """q [B, Pq, C]; cross(q + qpe) -> the cross-attention output."""
        if not self.cross_only:
            s = q + qpe
            q = self.norm1(q + self.dropout1(self.self_attn(s, s, s)))
        q = self.norm2(q + self.dropout2(cross(q + qpe)))
        q2 = self.linear2(self.dropout(self.activation(self.linear1(q))))
        return self.norm3(q + self.dropout3(q2))


class FFN(nn.Module):
    """Prediction heads of one decoder layer (transfusion_head.py:507-591): per head (ConvModule 1x1 + BN1d + ReLU)
    x (num_conv - 1), then Conv1d -> classes."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19, conv_cfg=dict(type='Conv1d'),
                 norm_cfg=dict(type='BN1d'), bias='auto', **kwargs):
        super(FFN, self).__init__()
        self.heads = heads
        self.init_bias = init_bias
        for head in self.heads:
            classes, num_conv = self.heads[head]
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(_ConvModule(c_in, head_conv, final_kernel, padding=final_kernel // 2, bias=bias,
                                          conv_cfg=conv_cfg, norm_cfg=norm_cfg))
                c_in = head_conv
            layers.append({'Conv1d': nn.Conv1d, 'Conv2d': nn.Conv2d}[conv_cfg['type']](
                c_in, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2, bias=True))
            self.__setattr__(head, nn.Sequential(*layers))

    def init_weights(self):
        for head in self.heads:
            if head == 'heatmap':
                self.__getattr__(head)[-1].bias.data.fill_(self.init_bias)

    def forward(self, x):
        return {head: self.__getattr__(head)(x) for head in self.heads}


class TransFusionBBoxCoder(object):
    """core/bbox/coders/transfusion_bbox_coder.py:8-128."""

    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None, score_threshold=None, code_size=8):
        self.pc_range, self.out_size_factor, self.voxel_size = pc_range, out_size_factor, voxel_size
        self.post_center_range, self.score_threshold, self.code_size = post_center_range, score_threshold, code_size

    def encode(self, dst_boxes):
        t = dst_boxes.new_zeros((dst_boxes.shape[0], self.code_size))
        t[:, 0] = (dst_boxes[:, 0] - self.pc_range[0]) / (self.out_size_factor * self.voxel_size[0])
        t[:, 1] = (dst_boxes[:, 1] - self.pc_range[1]) / (self.out_size_factor * self.voxel_size[1])
        t[:, 3:6] = dst_boxes[:, 3:6].log()
        t[:, 2] = dst_boxes[:, 2] + dst_boxes[:, 5] * 0.5
        t[:, 6], t[:, 7] = torch.sin(dst_boxes[:, 6]), torch.cos(dst_boxes[:, 6])
        if self.code_size == 10:
            t[:, 8:10] = dst_boxes[:, 7:]
        return t

    def decode(self, heatmap, rot, dim, center, height, vel, filter=False):
        """Plain torch decode (any device); the head's `get_bboxes` uses df3d_transfusion_decode instead."""
        final_scores, final_preds = heatmap.max(1)
        center = torch.stack([center[:, 0] * self.out_size_factor * self.voxel_size[0] + self.pc_range[0],
                              center[:, 1] * self.out_size_factor * self.voxel_size[1] + self.pc_range[1]], 1)
        dim = dim.exp()
        height = height - dim[:, 2:3, :] * 0.5
        rot = torch.atan2(rot[:, 0:1, :], rot[:, 1:2, :])
        parts = [center, height, dim, rot] + ([vel] if vel is not None else [])
        boxes = torch.cat(parts, dim=1).permute(0, 2, 1)
        B = heatmap.shape[0]
        if filter is False:
            return [dict(bboxes=boxes[i], scores=final_scores[i], labels=final_preds[i]) for i in range(B)]
        if self.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only support post_center_range is not None for now!')
        rng = _ops.device_constant([float(v) for v in self.post_center_range], boxes.dtype, heatmap.device)
        mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
        if self.score_threshold:
            mask &= final_scores > self.score_threshold
        return [dict(bboxes=boxes[i, mask[i]], scores=final_scores[i, mask[i]], labels=final_preds[i, mask[i]])
                for i in range(B)]


class _GaussianFocalFunction(torch.autograd.Function):
    """loss_heatmap on the device (df3d_gaussian_focal_loss); backward = the stored gradient x its normaliser."""

    @staticmethod
    def forward(ctx, logits, target, alpha, gamma, loss_weight, want_grad):
        out, grad = _ops.gaussian_focal_loss(logits.detach(), target, alpha, gamma, loss_weight, want_grad=want_grad)
        ctx.saved = (grad, out)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        grad, out = ctx.saved
        if grad is None:
            raise RuntimeError("loss_device ran without gradients enabled")
        return grad * (g * out[2]), None, None, None, None, None


class _QueryLossFunction(torch.autograd.Function):
    """Per-layer classification / box losses (df3d_tf_query_loss).  The kernel writes d(loss_cls_l) / d(class logits) and
    d(loss_bbox_l) / d(box codes) into disjoint columns of one buffer; backward scales each block by its upstream gradient."""

    @staticmethod
    def forward(ctx, rows, assigned, iou, K, code, C, gt, lab, off, step, pc_range, alpha, gamma, w_cls, w_bbox, pos_weight,
                code_weights, want_grad):
        out, grad = _ops.tf_query_loss(rows.detach(), assigned, iou, K, code, C, code, gt, lab, off, step, pc_range, alpha, gamma,
                                       w_cls, w_bbox, pos_weight, code_weights, want_grad=want_grad)
        ctx.saved = (grad, K, code, C)
        return out

    @staticmethod
    def backward(ctx, g):
        grad, K, code, C = ctx.saved
        if grad is None:
            raise RuntimeError("loss_device ran without gradients enabled")
        B, P_all, ld = grad.shape
        layers = P_all // K
        pair = g[:2 * layers].view(layers, 1, 2)                                 # (d cls, d bbox) per layer
        mult = torch.cat([pair[..., 1:2].expand(layers, K, code), pair[..., 0:1].expand(layers, K, C)], 2).reshape(1, P_all, ld)
        return (grad * mult,) + (None,) * 17


_EXEMPT = {'nuScenes': (8, 9), 'Waymo': (1, 2)}        # transfusion_head.py:856-861


@HEADS.register_module
@MM_HEADS.register_module
class TransFusionHead(nn.Module):
    def __init__(self, fuse_img=False, num_views=0, in_channels_img=64, out_size_factor_img=4, num_proposals=128,
                 auxiliary=True, in_channels=128 * 3, hidden_channel=128, num_classes=4, num_decoder_layers=3, num_heads=8,
                 learnable_query_pos=False, initialize_by_heatmap=False, nms_kernel_size=1, ffn_channel=256, dropout=0.1,
                 bn_momentum=0.1, activation='relu', common_heads=dict(), num_heatmap_convs=2,
                 conv_cfg=dict(type='Conv1d'), norm_cfg=dict(type='BN1d'), bias='auto', loss_cls=None, loss_iou=None,
                 loss_bbox=None, loss_heatmap=None, train_cfg=None, test_cfg=None, bbox_coder=None):
        super(TransFusionHead, self).__init__()
        if fuse_img:
            raise NotImplementedError("fuse_img=True (the camera decoder of TransFusion-LC) is not part of the "
                                      "3D-Dual-Fusion configs, which use the LiDAR-only head")
        self.num_classes = num_classes
        self.num_proposals = num_proposals
        self.auxiliary = auxiliary
        self.in_channels = in_channels
        self.num_heads = num_heads
        self.num_decoder_layers = num_decoder_layers
        self.bn_momentum = bn_momentum
        self.learnable_query_pos = learnable_query_pos
        self.initialize_by_heatmap = initialize_by_heatmap
        self.nms_kernel_size = nms_kernel_size
        if initialize_by_heatmap:
            assert learnable_query_pos is False, "initialized by heatmap is conflicting with learnable query position"
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.use_sigmoid_cls = (loss_cls or {}).get('use_sigmoid', False)
        if not self.use_sigmoid_cls:
            self.num_classes += 1
        coder = dict(bbox_coder or {})
        coder.pop('type', None)
        self.bbox_coder = TransFusionBBoxCoder(**coder) if coder else None
        self.loss_cls, self.loss_bbox = _tl.build_loss(loss_cls), _tl.build_loss(loss_bbox)
        self.loss_iou, self.loss_heatmap = _tl.build_loss(loss_iou), _tl.build_loss(loss_heatmap)
        self.sampling = False
        self.fuse_img = False
        self.shared_conv = nn.Conv2d(in_channels, hidden_channel, kernel_size=3, padding=1, bias=bool(bias))
        if initialize_by_heatmap:
            self.heatmap_head = nn.Sequential(
                _ConvModule(hidden_channel, hidden_channel, 3, padding=1, bias=bias, conv_cfg=dict(type='Conv2d'),
                            norm_cfg=dict(type='BN2d')),
                nn.Conv2d(hidden_channel, num_classes, kernel_size=3, padding=1, bias=bool(bias)))
            self.class_encoding = nn.Conv1d(num_classes, hidden_channel, 1)
        else:
            self.query_feat = nn.Parameter(torch.randn(1, hidden_channel, num_proposals))
            self.query_pos = nn.Parameter(torch.rand([1, num_proposals, 2]), requires_grad=learnable_query_pos)
        self.decoder = nn.ModuleList([
            TransformerDecoderLayer(hidden_channel, num_heads, ffn_channel, dropout, activation,
                                    self_posembed=PositionEmbeddingLearned(2, hidden_channel),
                                    cross_posembed=PositionEmbeddingLearned(2, hidden_channel))
            for _ in range(num_decoder_layers)])
        self.prediction_heads = nn.ModuleList()
        for _ in range(num_decoder_layers):
            heads = copy.deepcopy(common_heads)
            heads.update(dict(heatmap=(self.num_classes, num_heatmap_convs)))
            self.prediction_heads.append(FFN(hidden_channel, heads, conv_cfg=conv_cfg, norm_cfg=norm_cfg, bias=bias))
        self.init_weights()
        x_size = self.test_cfg['grid_size'][0] // self.test_cfg['out_size_factor']
        y_size = self.test_cfg['grid_size'][1] // self.test_cfg['out_size_factor']
        self.bev_pos = self.create_2D_grid(x_size, y_size)
        self.query_labels = None
        self._init_assigner_sampler()

    def _init_assigner_sampler(self):
        """transfusion_head.py:781-795."""
        if self.train_cfg is None:
            return
        self.bbox_sampler = _tl.PseudoSampler()
        assigner = self.train_cfg['assigner']
        self.bbox_assigner = ([_tl.build_assigner(a) for a in assigner] if isinstance(assigner, list)
                              else _tl.build_assigner(assigner))

    def create_2D_grid(self, x_size, y_size):
        """[1, x_size * y_size, 2]: entry i * y_size + j = (j + 0.5, i + 0.5) (transfusion_head.py:758-765)."""
        by, bx = torch.meshgrid(torch.linspace(0, x_size - 1, x_size), torch.linspace(0, y_size - 1, y_size), indexing="ij")
        return torch.stack([bx + 0.5, by + 0.5], 0).view(1, 2, -1).permute(0, 2, 1).contiguous()

    def init_weights(self):
        for m in self.decoder.parameters():
            if m.dim() > 1:
                nn.init.xavier_uniform_(m)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum

    def train(self, mode=True):
        self.__dict__.pop("_row_plan", None)
        self.__dict__.pop("_plan_tensors", None)
        return super(TransFusionHead, self).train(mode)

    # ------------------------------------------------------------------ plain torch path (training / CPU)
    def forward_reference(self, inputs, convs=None):
        """forward_single (transfusion_head.py:797-1030), LiDAR-only statements.  convs: callable that produces (shared-conv
        map, dense heat map) -- the training forward on the row kernels (`_train_convs_rows`); None = the torch modules."""
        B = inputs.shape[0]
        if convs is not None:
            lidar_feat, dense_heatmap = convs(inputs)
            flat = lidar_feat.flatten(2)                            # a view of the channels-last rows ([B, C, HW] strided)
        else:
            lidar_feat = self.shared_conv(inputs)
            flat = lidar_feat.view(B, lidar_feat.shape[1], -1)
        bev_pos = self.bev_pos.repeat(B, 1, 1).to(lidar_feat)
        if self.initialize_by_heatmap:
            if convs is None:
                dense_heatmap = self.heatmap_head(lidar_feat)
            heatmap = dense_heatmap.detach().sigmoid()
            pad = self.nms_kernel_size // 2
            local_max = torch.zeros_like(heatmap)
            inner = F.max_pool2d(heatmap, kernel_size=self.nms_kernel_size, stride=1, padding=0)
            local_max[:, :, pad:heatmap.shape[2] - pad, pad:heatmap.shape[3] - pad] = inner
            for c in _EXEMPT.get(self.test_cfg['dataset'], ()):
                local_max[:, c] = heatmap[:, c]
            heatmap = (heatmap * (heatmap == local_max)).reshape(B, heatmap.shape[1], -1)
            top = heatmap.reshape(B, -1).argsort(dim=-1, descending=True, stable=True)[..., :self.num_proposals]
            top_class = top // heatmap.shape[-1]
            top_index = top % heatmap.shape[-1]
            query_feat = flat.gather(index=top_index[:, None, :].expand(-1, flat.shape[1], -1), dim=-1)
            self.query_labels = top_class
            one_hot = F.one_hot(top_class, num_classes=self.num_classes).permute(0, 2, 1)
            query_feat = query_feat + self.class_encoding(one_hot.to(query_feat.dtype))
            query_pos = bev_pos.gather(index=top_index[:, :, None].expand(-1, -1, 2), dim=1)
        else:
            query_feat = self.query_feat.repeat(B, 1, 1)
            query_pos = self.query_pos.repeat(B, 1, 1).to(lidar_feat.device)
        ret_dicts = []
        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, flat, query_pos, bev_pos)
            res = self.prediction_heads[i](query_feat)
            res['center'] = res['center'] + query_pos.permute(0, 2, 1)
            ret_dicts.append(res)
            query_pos = res['center'].detach().clone().permute(0, 2, 1)
        if self.initialize_by_heatmap:
            ret_dicts[0]['query_heatmap_score'] = heatmap.gather(
                index=top_index[:, None, :].expand(-1, self.num_classes, -1), dim=-1)
            ret_dicts[0]['dense_heatmap'] = dense_heatmap
        return self._collect(ret_dicts)

    def _collect(self, ret_dicts):
        if self.auxiliary is False:
            return [ret_dicts[-1]]
        new_res = {}
        for key in ret_dicts[0].keys():
            if key not in ('dense_heatmap', 'dense_heatmap_old', 'query_heatmap_score'):
                new_res[key] = torch.cat([r[key] for r in ret_dicts], dim=-1) if len(ret_dicts) > 1 else ret_dicts[0][key]
            else:
                new_res[key] = ret_dicts[0][key]
        return [new_res]

    def _train_convs_rows(self, inputs):
        """shared_conv and the heat-map branch of a TRAINING forward on the row kernels (the neck's `_train_stack`: every 3 x 3
        convolution is `SparseConvFunction` over the full neighbour table of the BEV grid -- forward, input gradient and filter
        gradient on csrc/spconv_split.hip / spconv_bwd.hip --, BatchNorm over the pixel rows on csrc/bnrows.hip) instead of
        MIOpen's fp32 convolutions (5.2 ms of the 68 ms TransFusion training step at bs 4).  -> (lidar_feat [B, C, H, W] as a
        view of channels-last rows, dense_heatmap [B, classes, H, W])."""
        from .necks import RPN, _rows_of, _train_stack
        from .spconv.conv import SparseConvFunction
        B, _, H, W = inputs.shape
        tables = self.__dict__.setdefault("_train_tables", {})
        rows, _ = _rows_of(inputs)
        key = (B, H, W)
        if key not in tables:
            tables[key] = _ops.conv2d_neighbors(B, H, W, 3, 3, 1, 1, False, rows.device)[0]
        nbr = tables[key]
        sc = self.shared_conv
        hm0, hm1 = self.heatmap_head[0], self.heatmap_head[1]
        C = hm1.out_channels
        with _ops.precision("split"):          # (the head's own convolutions stay fp32-grade in the bf16 mode, as in inference)
            feat = SparseConvFunction.apply(rows.contiguous(), sc.weight.permute(2, 3, 1, 0), sc.bias, nbr, nbr.shape[1], True, None)
            mid, _, _ = _train_stack([(hm0.conv, hm0.bn, True, 1)], feat, B, H, W,
                                     self.__dict__.setdefault("_train_stack_tables", {}))
            # the class maps as 32 output columns (zero filters behind the real ones: the narrowest block of the row kernel;
            # autograd slices the gradient of the padding away)
            w = F.pad(hm1.weight.permute(2, 3, 1, 0), (0, 32 - C))
            b = F.pad(hm1.bias, (0, 32 - C)) if hm1.bias is not None else None
            heat = SparseConvFunction.apply(mid.contiguous(), w, b, nbr, nbr.shape[1], True, None)[:, :C]
        to_map = lambda r: r.reshape(B, H, W, -1).permute(0, 3, 1, 2)                       # noqa: E731
        return to_map(feat), to_map(heat)

    def forward_single(self, inputs, img_inputs=None, img_metas=None):
        if (self.training or torch.is_grad_enabled() or not inputs.is_cuda or inputs.dtype != torch.float32
                or _ops.CONV_PRECISION == "fp32" or not self._row_kernels_fit(inputs)):
            if (torch.is_grad_enabled() and inputs.is_cuda and inputs.dtype == torch.float32 and self._row_kernels_fit(inputs)
                    and os.environ.get("DF3D_TFHEAD_TRAIN_ROWS", "1") == "1"):
                return self.forward_reference(inputs, convs=self._train_convs_rows)
            return self.forward_reference(inputs)
        # the head's own convolutions stay split precision (fp32-grade) in the bf16 mode of the backbone / neck
        return self.forward_rows(inputs)

    def forward(self, feats, img_feats=None, img_metas=None):
        """feats: list with ONE BEV map [B, C, H, W] (or the tensor).  Returns ([dict],) like the reference's
        multi_apply over levels (transfusion_head.py:1032-1046)."""
        if torch.is_tensor(feats):
            feats = [feats]
        assert len(feats) == 1, "only support one level features."
        return ([self.forward_single(feats[0], None, img_metas)[0]],)

    # ------------------------------------------------------------------ device path
    def _row_kernels_fit(self, x):
        hm = getattr(self, "heatmap_head", None)
        return (self.initialize_by_heatmap and self.shared_conv.in_channels == 512 and self.shared_conv.out_channels == 128
                and self.shared_conv.bias is not None and self.num_classes <= 32 and self.nms_kernel_size >= 3
                and self.nms_kernel_size % 2 == 1 and x.shape[2] * x.shape[3] == self.bev_pos.shape[1]
                and hm is not None and hm[1].bias is not None and hasattr(hm[0], "bn"))

    @staticmethod
    def _filters(conv, pad_to=None):
        w = conv.weight.detach().float().permute(2, 3, 1, 0).reshape(9, conv.in_channels, conv.out_channels)
        if pad_to is not None and pad_to > conv.out_channels:
            w = torch.cat([w, w.new_zeros(9, conv.in_channels, pad_to - conv.out_channels)], 2)
        return w.contiguous()

    @staticmethod
    def _fold(bn):
        scale = bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)
        return scale.contiguous(), (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()

    def _plan(self):
        params = self.__dict__.get("_plan_tensors")
        if params is None:                            # collected once (train() drops it): parameters() walks the module tree
            params = self.__dict__["_plan_tensors"] = [p for p in self.parameters()] + [b for b in self.buffers()]
        key = tuple((p.data_ptr(), p._version) for p in params)
        plan = self.__dict__.get("_row_plan")
        if plan is not None and plan["key"] == key:
            return plan
        dev = self.shared_conv.weight.device
        hm0, hm1 = self.heatmap_head[0], self.heatmap_head[1]
        C = self.num_classes
        m_scale, m_shift = self._fold(hm0.bn)
        plan = dict(key=key, nbr={}, kv_const={},
                    shared=_ops.conv_pack_weights(self._filters(self.shared_conv)),
                    s_bias=self.shared_conv.bias.detach().float().contiguous(),
                    mid=_ops.conv_pack_weights(self._filters(hm0.conv)),
                    m_bias=hm0.conv.bias.detach().float().contiguous() if hm0.conv.bias is not None else None,
                    m_scale=m_scale, m_shift=m_shift,
                    fin=_ops.conv_pack_weights(self._filters(hm1, 32)),
                    f_bias=torch.cat([hm1.bias.detach().float(), torch.zeros(32 - C, device=dev)]).contiguous(),
                    cols=torch.tensor([[0, C]], dtype=torch.int32, device=dev), width=(C + 7) // 8 * 8,
                    cls_w=self.class_encoding.weight.detach().float().reshape(-1, C).contiguous(),
                    cls_b=self.class_encoding.bias.detach().float().contiguous(), layers=[])
        for i in range(self.num_decoder_layers):
            dec, ffn = self.decoder[i], self.prediction_heads[i]
            E = dec.multihead_attn.embed_dim
            w, b = dec.multihead_attn.in_proj_weight.detach().float(), dec.multihead_attn.in_proj_bias.detach().float()
            w1, b1, blocks, b2, layout, c0 = [], [], [], [], [], 0
            for head in ffn.heads:
                fc = getattr(ffn, head)
                if len(fc) != 2:
                    raise NotImplementedError("prediction heads with num_conv != 2")
                a, s = self._fold(fc[0].bn) if hasattr(fc[0], "bn") else (None, None)
                cw = fc[0].conv.weight.detach().float().reshape(fc[0].conv.out_channels, -1)
                cb = fc[0].conv.bias.detach().float() if fc[0].conv.bias is not None else cw.new_zeros(cw.shape[0])
                w1.append(cw * a[:, None] if a is not None else cw)
                b1.append(cb * a + s if a is not None else cb)
                blocks.append(fc[1].weight.detach().float().reshape(fc[1].out_channels, -1))
                b2.append(fc[1].bias.detach().float())
                layout.append((head, c0, fc[1].out_channels))
                c0 += fc[1].out_channels
            L = dict(wq=w[:E].contiguous(), bq=b[:E].contiguous(), wkv_t=w[E:].t().contiguous(),
                     bkv=b[E:].contiguous(), w1=torch.cat(w1).contiguous(), b1=torch.cat(b1).contiguous(),
                     w2=torch.block_diag(*blocks).contiguous(), b2=torch.cat(b2).contiguous(), layout=layout)
            # row form of the layer's small operators (200 x 128 activations): transposed weights for addmm, the
            # query position embedding's BatchNorm folded into its first 1x1 convolution
            t = lambda m: m.detach().float().t().contiguous()
            v = lambda m: m.detach().float().contiguous()
            pe = dec.self_posembed.position_embedding_head
            a, sft = self._fold(pe[1])
            sa = getattr(dec, "self_attn", None)
            L.update(wq_t=t(w[:E]), pe_w1t=(pe[0].weight.detach().float().reshape(E, -1) * a[:, None]).t().contiguous(),
                     pe_b1=(pe[0].bias.detach().float() * a + sft).contiguous(),
                     pe_w2t=t(pe[3].weight.reshape(pe[3].out_channels, -1)), pe_b2=v(pe[3].bias),
                     sa_wt=t(sa.in_proj_weight) if sa is not None else None,
                     sa_b=v(sa.in_proj_bias) if sa is not None else None,
                     sa_owt=t(sa.out_proj.weight) if sa is not None else None,
                     sa_ob=v(sa.out_proj.bias) if sa is not None else None,
                     ca_owt=t(dec.multihead_attn.out_proj.weight), ca_ob=v(dec.multihead_attn.out_proj.bias),
                     l1_wt=t(dec.linear1.weight), l1_b=v(dec.linear1.bias), l2_wt=t(dec.linear2.weight),
                     l2_b=v(dec.linear2.bias))
            plan["layers"].append(L)
        self.__dict__["_row_plan"] = plan
        return plan

    def _kv_const(self, plan, i, B, dev):
        """rows of  cross_posembed(bev_pos) @ W_kv^T + b_kv  [B*H*W, 2E]: the additive term of the key / value GEMM."""
        if (i, B) not in plan["kv_const"]:
            L = plan["layers"][i]
            kpe = self.decoder[i].cross_posembed(self.bev_pos.to(dev))[0].t()               # [HW, E]
            plan["kv_const"][(i, B)] = torch.addmm(L["bkv"], kpe, L["wkv_t"]).repeat(B, 1).contiguous()
        return plan["kv_const"][(i, B)]

    def _decoder_rows(self, dec, L, q, qpos, kv, B):
        """TransformerDecoderLayer.forward (transfusion_head.py:82-122) in eval mode on [B*K, E] rows: the same
        operators as `TransformerDecoderLayer.finish`, with fused-bias GEMMs, one add + LayerNorm launch per residual
        and both attentions through df3d_cross_attention."""
        E, mha = q.shape[1], dec.multihead_attn
        if mha.head_dim != 16 or E != mha.embed_dim:
            K = q.shape[0] // B
            qpe = dec.self_posembed(qpos.view(B, K, 2)).transpose(1, 2)
            kv3 = kv.view(B, -1, 2 * E)
            return dec.finish(q.view(B, K, E), qpe, lambda qq: mha.attend(F.linear(qq, L["wq"], L["bq"]), kv3[..., :E],
                                                                          kv3[..., E:])).reshape(B * K, E)
        scale = mha.head_dim ** -0.5
        qpe = torch.addmm(L["pe_b2"], torch.relu_(torch.addmm(L["pe_b1"], qpos, L["pe_w1t"])), L["pe_w2t"])
        if not dec.cross_only:
            qkv = torch.addmm(L["sa_b"], q + qpe, L["sa_wt"])                              # [B*K, 3E]
            o = _ops.cross_attention(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], B, mha.num_heads, scale)
            q = _ops.add_layernorm(q, torch.addmm(L["sa_ob"], o, L["sa_owt"]), dec.norm1.weight, dec.norm1.bias, dec.norm1.eps)
        qp = torch.addmm(L["bq"], q + qpe, L["wq_t"])
        o = _ops.cross_attention(qp, kv[:, :E], kv[:, E:], B, mha.num_heads, scale)
        q = _ops.add_layernorm(q, torch.addmm(L["ca_ob"], o, L["ca_owt"]), dec.norm2.weight, dec.norm2.bias, dec.norm2.eps)
        f = torch.addmm(L["l2_b"], dec.activation(torch.addmm(L["l1_b"], q, L["l1_wt"])), L["l2_wt"])
        return _ops.add_layernorm(q, f, dec.norm3.weight, dec.norm3.bias, dec.norm3.eps)

    @torch.no_grad()
    def forward_rows(self, x):
        from .necks import _rows_of
        plan = self._plan()
        B, _, H, W = x.shape
        n, C, K, dev = B * H * W, self.num_classes, self.num_proposals, x.device
        rows, split = _rows_of(x)
        if split is None or split.dtype != torch.uint8:     # no hi/lo rows cached (bf16 mode of the neck caches bf16 rows)
            split = _ops.split_rows(rows.contiguous())
        if (B, H, W) not in plan["nbr"]:
            plan["nbr"][(B, H, W)] = _ops.conv2d_neighbors(B, H, W, 3, 3, 1, 1, False, dev)[0]
        nbr = plan["nbr"][(B, H, W)]
        feat, s1 = _ops.conv_rows_split(split, 512, 0, plan["shared"], 128, 1, nbr, n, plan["s_bias"], None, None,
                                        relu=False, want_out=True, want_split=True)
        _, s2 = _ops.conv_rows_split(s1, 128, 0, plan["mid"], 128, 1, nbr, n, plan["m_bias"], plan["m_scale"],
                                     plan["m_shift"], relu=True, want_out=False, want_split=True)
        heat, _ = _ops.conv_rows_split(s2, 128, 0, plan["fin"], 32, 1, nbr, n, plan["f_bias"], None, None, relu=False,
                                       out_channels=plan["width"], out_cols=plan["cols"])
        top_class, top_pixel, qscore, query_pos, qf = _ops.heatmap_proposals(
            heat[:, :C], B, C, H, W, self.nms_kernel_size, _EXEMPT.get(self.test_cfg['dataset'], ()), K, feat,
            plan["cls_w"], plan["cls_b"])
        self.query_labels = top_class.long()
        E = feat.shape[1]
        ret_dicts = []
        for i in range(self.num_decoder_layers):
            dec, L = self.decoder[i], plan["layers"][i]
            kv = torch.addmm(self._kv_const(plan, i, B, dev), feat, L["wkv_t"])             # [B*H*W, 2E]
            qf = self._decoder_rows(dec, L, qf.reshape(B * K, E), query_pos.reshape(B * K, 2), kv, B).view(B, K, E)
            out = torch.addmm(L["b2"], torch.relu_(torch.addmm(L["b1"], qf.reshape(B * K, E), L["w1"].t())), L["w2"].t())
            out = out.view(B, K, -1)
            res = {}
            for head, c0, k in L["layout"]:
                if head == 'center':
                    out[..., c0:c0 + k] += query_pos
                res[head] = out[..., c0:c0 + k].permute(0, 2, 1)
            ret_dicts.append(res)
            query_pos = res['center'].permute(0, 2, 1).clone()
        ret_dicts[0]['query_heatmap_score'] = qscore
        ret_dicts[0]['dense_heatmap'] = heat[:, :C].view(B, H, W, C).permute(0, 3, 1, 2)
        return self._collect(ret_dicts)

    # ------------------------------------------------------------------ boxes
    def get_targets(self, gt_bboxes_3d, gt_labels_3d, preds_dict):
        """transfusion_head.py:1048-1086: per-sample targets concatenated over the batch.  Returns labels [B, P],
        label_weights [B, P], bbox_targets [B, P, code], bbox_weights [B, P, code], ious [B, P], num_pos, matched_ious
        (mean over samples) and, with heat-map initialisation, the dense heat-map target [B, C, H, W]."""
        per_sample = []
        for b in range(len(gt_bboxes_3d)):
            one = {k: v[b:b + 1] for k, v in preds_dict[0].items()}
            per_sample.append(self.get_targets_single(gt_bboxes_3d[b], gt_labels_3d[b], one, b))
        cols = list(zip(*per_sample))
        out = [torch.cat(cols[i], dim=0) for i in range(5)] + [int(sum(cols[5])), float(sum(cols[6]) / max(len(cols[6]), 1))]
        if self.initialize_by_heatmap:
            out.append(torch.cat(cols[7], dim=0))
        return tuple(out)

    def _gt_tensor(self, gt_bboxes_3d, device):
        t = gt_bboxes_3d.tensor if hasattr(gt_bboxes_3d, "tensor") else gt_bboxes_3d
        return t.to(device)

    def get_targets_single(self, gt_bboxes_3d, gt_labels_3d, preds_dict, batch_idx):
        """transfusion_head.py:1088-1216 for one sample (HungarianAssigner3D + PseudoSampler)."""
        if self.train_cfg is None:
            raise RuntimeError("TransFusionHead needs train_cfg for target assignment")
        if self.train_cfg['assigner']['type'] != 'HungarianAssigner3D':
            raise NotImplementedError("only HungarianAssigner3D (the 3D-Dual-Fusion configs' assigner)")
        P_all = preds_dict['center'].shape[-1]
        dev = preds_dict['center'].device
        d = {k: preds_dict[k].detach().clone() for k in ('heatmap', 'center', 'height', 'dim', 'rot')}
        vel = preds_dict['vel'].detach().clone() if 'vel' in preds_dict else None
        boxes = self.bbox_coder.decode(d['heatmap'], d['rot'], d['dim'], d['center'], d['height'], vel)[0]['bboxes']
        gt = self._gt_tensor(gt_bboxes_3d, dev)
        gt_labels_3d = gt_labels_3d.to(dev)
        layers = self.num_decoder_layers if self.auxiliary else 1
        K = self.num_proposals
        results = [self.bbox_assigner.assign(boxes[K * i:K * (i + 1)], gt, gt_labels_3d, d['heatmap'][..., K * i:K * (i + 1)],
                                             self.train_cfg) for i in range(layers)]
        gt_inds = torch.cat([r.gt_inds for r in results])
        # the reference concatenates `max_overlaps` and so cannot take a frame without ground truth (None); zeros here
        ious = torch.cat([r.max_overlaps if r.max_overlaps is not None else boxes.new_zeros(K) for r in results])
        sampled = self.bbox_sampler.sample(_tl.AssignResult(sum(r.num_gts for r in results), gt_inds, ious,
                                                            torch.cat([r.labels for r in results])), boxes, gt)
        pos, neg = sampled.pos_inds, sampled.neg_inds
        assert len(pos) + len(neg) == P_all
        code = self.bbox_coder.code_size
        bbox_targets, bbox_weights = boxes.new_zeros((P_all, code)), boxes.new_zeros((P_all, code))
        ious = torch.clamp(ious, min=0.0, max=1.0)
        labels = boxes.new_zeros(P_all, dtype=torch.long)
        label_weights = boxes.new_zeros(P_all, dtype=torch.long)
        if gt_labels_3d is not None:
            labels += self.num_classes
        if len(pos) > 0:
            bbox_targets[pos, :] = self.bbox_coder.encode(sampled.pos_gt_bboxes)
            bbox_weights[pos, :] = 1.0
            labels[pos] = gt_labels_3d[sampled.pos_assigned_gt_inds] if gt_labels_3d is not None else 1
            pw = self.train_cfg['pos_weight']
            label_weights[pos] = 1.0 if pw <= 0 else pw
        if len(neg) > 0:
            label_weights[neg] = 1.0
        mean_iou = float(ious[pos].sum() / max(len(pos), 1))
        ret = [labels[None], label_weights[None], bbox_targets[None], bbox_weights[None], ious[None], int(pos.shape[0]), mean_iou]
        if self.initialize_by_heatmap:
            ret.append(self.heatmap_targets([gt], [gt_labels_3d], dev))
        return tuple(ret)

    def _splat_geometry(self, gt):
        """Per ground-truth box (x, y, z_bottom, w, l, h, ...): integer centre pixel and radius of its Gaussian
        (transfusion_head.py:1186-1207), element-wise in fp32; radius < 0 marks a box that is not drawn."""
        cfg = self.train_cfg
        osf = cfg['out_size_factor']
        vs = gt.new_tensor(cfg['voxel_size'])
        pc = gt.new_tensor(cfg['point_cloud_range'])
        width, length = gt[:, 3] / vs[0] / osf, gt[:, 4] / vs[1] / osf
        ok = (width > 0) & (length > 0)
        radius = _tl.gaussian_radius((length, width), min_overlap=cfg['gaussian_overlap'])
        radius = torch.clamp(torch.nan_to_num(radius, nan=0.0).to(torch.int32), min=int(cfg['min_radius']))
        cx = ((gt[:, 0] - pc[0]) / vs[0] / osf).to(torch.int32)
        cy = ((gt[:, 1] - pc[1]) / vs[1] / osf).to(torch.int32)
        return cx, cy, torch.where(ok, radius, torch.full_like(radius, -1))

    def heatmap_targets(self, gt_list, label_list, device):
        """Dense heat-map targets [B, C, y_len, x_len] of a list of per-sample ground truth tensors."""
        cfg = self.train_cfg
        fx, fy = [int(g) // cfg['out_size_factor'] for g in cfg['grid_size'][:2]]
        heatmap = torch.zeros((len(gt_list), self.num_classes, fy, fx), dtype=torch.float32, device=device)
        for b, (gt, lab) in enumerate(zip(gt_list, label_list)):
            if gt.shape[0] == 0:
                continue
            cx, cy, radius = [t.tolist() for t in self._splat_geometry(gt.float())]
            for i, c in enumerate(lab.tolist()):
                if radius[i] >= 0:
                    _tl.draw_heatmap_gaussian(heatmap[b, c], (cx[i], cy[i]), radius[i])
        return heatmap

    def loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, **kwargs):
        """transfusion_head.py:1218-1283: dict(loss_heatmap, layer_{i}_loss_cls, layer_{i}_loss_bbox, matched_ious).
        Like the reference this turns `preds_dicts[0][0]['dense_heatmap']` into clamped probabilities IN PLACE."""
        targets = self.get_targets(gt_bboxes_3d, gt_labels_3d, preds_dicts[0])
        labels, label_weights, bbox_targets, bbox_weights, ious, num_pos, matched_ious = targets[:7]
        if hasattr(self, 'on_the_image_mask'):
            label_weights = label_weights * self.on_the_image_mask
            bbox_weights = bbox_weights * self.on_the_image_mask[:, :, None]
            num_pos = bbox_weights.max(-1).values.sum()
        p = preds_dicts[0][0]
        losses = dict()
        if self.initialize_by_heatmap:
            heatmap = targets[7]
            losses['loss_heatmap'] = self.loss_heatmap(_tl.clip_sigmoid(p['dense_heatmap']), heatmap,
                                                       avg_factor=max(heatmap.eq(1).float().sum().item(), 1))
        K = self.num_proposals
        code_weights = self.train_cfg.get('code_weights', None)
        layers = self.num_decoder_layers if self.auxiliary else 1
        for i in range(layers):
            last = i == self.num_decoder_layers - 1 or (i == 0 and self.auxiliary is False)
            prefix = 'layer_-1' if last else 'layer_%d' % i
            sl = slice(i * K, (i + 1) * K)
            scores = p['heatmap'][..., sl].permute(0, 2, 1).reshape(-1, self.num_classes)
            loss_cls = self.loss_cls(scores, labels[..., sl].reshape(-1), label_weights[..., sl].reshape(-1),
                                     avg_factor=max(num_pos, 1))
            parts = [p[k][..., sl] for k in ('center', 'height', 'dim', 'rot') + (('vel',) if 'vel' in p else ())]
            boxes = torch.cat(parts, dim=1).permute(0, 2, 1)
            reg_weights = bbox_weights[:, sl, :] * bbox_weights.new_tensor(code_weights)
            losses[prefix + '_loss_cls'] = loss_cls
            losses[prefix + '_loss_bbox'] = self.loss_bbox(boxes, bbox_targets[:, sl, :], reg_weights, avg_factor=max(num_pos, 1))
        losses['matched_ious'] = loss_cls.new_tensor(matched_ious)
        return losses

    # ------------------------------------------------------------------ losses on the device (csrc/tfloss.hip)
    def _pack_gt(self, gt_bboxes_3d, gt_labels_3d, dev):
        """Ground truth of a batch as one [G, D] fp32 tensor + labels [G] i32 + offsets [B + 1] i32 on the device, and the
        per-sample counts on the host (they come from the host: no synchronisation)."""
        boxes = [(g.tensor if hasattr(g, "tensor") else g).float() for g in gt_bboxes_3d]
        counts = [int(t.shape[0]) for t in boxes]
        dim = max([t.shape[1] for t in boxes] + [7])
        gt = torch.cat([t.reshape(-1, dim) for t in boxes]) if sum(counts) else torch.zeros((0, dim))
        lab = torch.cat([l.reshape(-1) for l in gt_labels_3d]).to(torch.int32) if sum(counts) else torch.zeros((0,), dtype=torch.int32)
        import numpy as np
        off = _ops.device_constant(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), torch.int32, dev)   # cached by value
        nb = gt.device.type == "cpu"
        return (gt.to(dev, non_blocking=nb).contiguous(), lab.to(dev, non_blocking=nb).contiguous(), off, counts)

    def _match_cfg(self):
        a, c = self.bbox_assigner, self.bbox_coder
        if not isinstance(a.cls_cost, _tl.FocalLossCost) or not isinstance(a.reg_cost, _tl.BBoxBEVL1Cost):
            raise NotImplementedError("the device matcher evaluates FocalLossCost + BBoxBEVL1Cost + IoU3DCost")
        return dict(out_size_factor=c.out_size_factor, voxel_size=c.voxel_size, pc_range=c.pc_range,
                    point_cloud_range=self.train_cfg['point_cloud_range'], cls_weight=a.cls_cost.weight,
                    cls_alpha=a.cls_cost.alpha, cls_gamma=a.cls_cost.gamma, cls_eps=a.cls_cost.eps,
                    reg_weight=a.reg_cost.weight, iou_weight=a.iou_cost.weight)

    def loss_device(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, **kwargs):
        """`loss` with every tensor operation on the device in seven launches for the whole batch: matching costs (decode +
        focal cost + BEV-centre cost + rotated 3-D IoU) -> host `linear_sum_assignment` per sample and decoder layer, as the
        reference -> query losses; Gaussian heat-map targets and their focal loss run while the host matches.  Differentiable
        (the kernels also write the gradients; `torch.is_grad_enabled()` decides).  Same keys / values as `loss`; unlike it
        the dense heat-map prediction is left untouched."""
        import numpy as np
        if self.train_cfg is None:
            raise RuntimeError("TransFusionHead needs train_cfg for target assignment")
        p = preds_dicts[0][0]
        dev = p['center'].device
        if dev.type != "cuda":
            raise _ops._lib.Df3dError("loss_device needs the predictions on the GPU (got %s)" % dev)
        names = ('center', 'height', 'dim', 'rot') + (('vel',) if 'vel' in p else ())
        code = self.bbox_coder.code_size
        rows = torch.cat([p[k] for k in names] + [p['heatmap']], 1).float().permute(0, 2, 1).contiguous()   # [B, P_all, code + C]
        if rows.shape[2] != code + self.num_classes:
            raise ValueError("prediction heads do not match bbox_coder.code_size = %d" % code)
        B, P_all, _ = rows.shape
        K, C = self.num_proposals, self.num_classes
        layers = self.num_decoder_layers if self.auxiliary else 1
        assert P_all == layers * K
        gt, lab, off, counts = self._pack_gt(gt_bboxes_3d, gt_labels_3d, dev)
        gmax = max(counts + [1])
        if sum(counts) == 0:
            # a batch in which NO sample has ground truth (accepted here): the kernels still want
            # non-null tables for gmax = 1 -- one zero row that no sample's [off[b], off[b+1]) range refers to
            gt, lab = gt.new_zeros((1, gt.shape[1])), lab.new_zeros((1,))
        want_grad = torch.is_grad_enabled() and (rows.requires_grad or p['dense_heatmap'].requires_grad)
        cfg = self.train_cfg
        cost, iou, _ = _ops.tf_match_cost(rows.detach(), code, C, gt, lab, off, gmax, **self._match_cfg())
        host = torch.empty(cost.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(cost, non_blocking=True)
        flag = None
        if _ops.CONV_PRECISION == "split":
            # the range flag of the fp16 operand format rides on the cost matrix's copy (no extra host wait): every kernel of
            # this forward is in front of it
            flag = torch.empty((1,), dtype=torch.int32, pin_memory=True)
            flag.copy_(_ops.overflow_word(), non_blocking=True)
        copied = torch.cuda.Event()
        copied.record()
        losses = dict()
        if self.initialize_by_heatmap:                      # independent of the matching: overlaps the host's work
            fx, fy = [int(g) // cfg['out_size_factor'] for g in cfg['grid_size'][:2]]
            target = _ops.draw_heatmap_gaussian(gt, lab, off, B, C, fy, fx, cfg['voxel_size'], cfg['out_size_factor'],
                                                cfg['point_cloud_range'], cfg['gaussian_overlap'], cfg['min_radius'])
            lh = self.loss_heatmap
            losses['loss_heatmap'] = _GaussianFocalFunction.apply(p['dense_heatmap'], target, lh.alpha, lh.gamma, lh.loss_weight,
                                                                 want_grad)
        copied.synchronize()
        if flag is not None:
            _ops.RANGE_STATS["range_checks"] += 1
            _ops.raise_if_range_flag(int(flag[0]))
        cost_np = host.numpy()
        assigned = np.full((B, P_all), -1, np.int32)
        start = 0
        for b, n in enumerate(counts):
            if n:
                for l in range(layers):
                    r, c = _tl.HungarianAssigner3D.match(cost_np[b, l * K:(l + 1) * K, :n])
                    assigned[b, l * K + np.asarray(r)] = start + np.asarray(c)
            start += n
        assigned = torch.from_numpy(assigned).to(dev, non_blocking=True)
        c = self.bbox_coder
        step = [float(c.out_size_factor * c.voxel_size[0]), float(c.out_size_factor * c.voxel_size[1])]
        out = _QueryLossFunction.apply(rows, assigned, iou, K, code, C, gt, lab, off, step, list(c.pc_range), self.loss_cls.alpha,
                                       self.loss_cls.gamma, self.loss_cls.loss_weight, self.loss_bbox.loss_weight,
                                       cfg['pos_weight'], list(cfg.get('code_weights', None) or [1.0] * code), want_grad)
        for i in range(layers):
            last = i == self.num_decoder_layers - 1 or (i == 0 and self.auxiliary is False)
            prefix = 'layer_-1' if last else 'layer_%d' % i
            losses[prefix + '_loss_cls'], losses[prefix + '_loss_bbox'] = out[2 * i], out[2 * i + 1]
        losses['matched_ious'] = out[2 * layers + 1].detach()
        self.__dict__['_last_num_pos'] = out[2 * layers].detach()
        return losses

    def get_bboxes_device(self, preds_dicts):
        """Device-resident result of get_bboxes with nms_type=None: (boxes [B, K, 7|9], scores [B, K], labels [B, K] i32,
        counts [B] i32); the first counts[b] entries of sample b are valid, in proposal order."""
        p = preds_dicts[0][0]
        K, C = self.num_proposals, self.num_classes
        B = p['heatmap'].shape[0]
        heads = {}
        for k in ('heatmap', 'center', 'height', 'dim', 'rot', 'vel'):
            if k in p:
                heads[k] = p[k][..., -K:].permute(0, 2, 1).reshape(B * K, -1).float()
        c = self.bbox_coder
        return _ops.transfusion_decode(heads, p['query_heatmap_score'].float().contiguous(),
                                       self.query_labels.to(torch.int32).contiguous(), B, K, C, c.out_size_factor,
                                       c.voxel_size, c.pc_range, c.post_center_range, c.score_threshold)

    @torch.no_grad()
    def get_bboxes(self, preds_dicts, img_metas=None, img=None, rescale=False, for_roi=False):
        """transfusion_head.py:1285-1376.  Returns one [boxes, scores, labels] triple per sample (the reference
        asserts a single sample and returns the one-element list)."""
        p = preds_dicts[0][0]
        nms_type = (self.test_cfg or {}).get('nms_type')
        if self.bbox_coder.post_center_range is None:
            raise NotImplementedError('Need to reorganize output as a batch, only support post_center_range is not None for now!')
        if p['heatmap'].is_cuda:
            boxes, scores, labels, counts = self.get_bboxes_device(preds_dicts)
            counts = _ops.read_with_range_flag(counts)          # (+ the range flag of the fp16 operand format)
            dets = [(boxes[b, :n], scores[b, :n], labels[b, :n].long()) for b, n in enumerate(counts)]
        else:
            K = self.num_proposals
            one_hot = F.one_hot(self.query_labels, num_classes=self.num_classes).permute(0, 2, 1)
            score = p['heatmap'][..., -K:].sigmoid() * p['query_heatmap_score'] * one_hot
            t = self.bbox_coder.decode(score, p['rot'][..., -K:], p['dim'][..., -K:], p['center'][..., -K:],
                                       p['height'][..., -K:], p['vel'][..., -K:] if 'vel' in p else None, filter=True)
            dets = [(d['bboxes'], d['scores'], d['labels']) for d in t]
        if nms_type is not None:
            dets = [self._task_nms(*d, nms_type) for d in dets]
        wrap = None
        if img_metas and isinstance(img_metas[0], dict) and callable(img_metas[0].get('box_type_3d')):
            wrap = img_metas[0]['box_type_3d']
        return [[wrap(b, box_dim=b.shape[-1]) if wrap else b, s, l.int()] for b, s, l in dets]

    def _task_nms(self, boxes3d, scores, labels, nms_type):
        """Per-task NMS of transfusion_head.py:1313-1357 (circle NMS on the device; task radii of the reference)."""
        from .iou3d_nms import circle_nms
        if nms_type != 'circle':
            raise NotImplementedError("nms_type=%r needs mmdet3d's box classes; the 3D-DF config uses None" % (nms_type,))
        if self.test_cfg['dataset'] == 'nuScenes':
            tasks = [(list(range(8)), -1), ([8], 0.175), ([9], 0.175)]
        elif self.test_cfg['dataset'] == 'Waymo':
            tasks = [([0], 0.7), ([1], 0.7), ([2], 0.7)]
        else:
            tasks = []
        keep = torch.zeros_like(scores, dtype=torch.bool)
        for classes, radius in tasks:
            m = torch.zeros_like(keep)
            for c in classes:
                m |= labels == c
            idx = torch.where(m)[0]
            if radius > 0 and idx.numel():
                idx = idx[circle_nms(torch.cat([boxes3d[idx][:, :2], scores[idx, None]], 1), radius).long()]
            keep[idx] = True
        return boxes3d[keep], scores[keep], labels[keep]
