"""Detection head of the CenterPoint tree: `CenterHead` / `SepHead` with the reference's
constructor arguments and parameter names (CP/det3d/models/bbox_heads/center_head.py:66-110,165-243), and
`predict` = the reference's decode + per-sample post-processing (center_head.py:302-501) as ONE device call
(`df3d_centerhead_predict`: keys -> radix sort -> gather -> NMS bit matrix -> on-device greedy reduction -> select)
instead of ~30 launches and three host round trips per (task, sample).  Forward-only: `loss` is a training row
(SURVEY.md section 8f row 4)."""
import copy

import torch
from torch import nn

from . import ops as _ops
from .registry import HEADS


class SepHead(nn.Module):
    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, bn=False, init_bias=-2.19, **kwargs):
        super(SepHead, self).__init__(**kwargs)
        self.heads = heads
        for head in self.heads:
            classes, num_conv = self.heads[head]
            mods = []
            for _ in range(num_conv - 1):
                mods.append(nn.Conv2d(in_channels, head_conv, kernel_size=final_kernel, stride=1,
                                      padding=final_kernel // 2, bias=True))
                if bn:
                    mods.append(nn.BatchNorm2d(head_conv))
                mods.append(nn.ReLU())
            mods.append(nn.Conv2d(head_conv, classes, kernel_size=final_kernel, stride=1, padding=final_kernel // 2,
                                  bias=True))
            fc = nn.Sequential(*mods)
            if 'hm' in head:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(head, fc)

    def forward(self, x):
        return {head: self.__getattr__(head)(x) for head in self.heads}


@HEADS.register_module
class CenterHead(nn.Module):
    def __init__(self, in_channels=[128, ], tasks=[], dataset='nuscenes', weight=0.25, code_weights=[],
                 common_heads=dict(), logger=None, init_bias=-2.19, share_conv_channel=64, num_hm_conv=2,
                 dcn_head=False):
        super(CenterHead, self).__init__()
        if dcn_head:
            raise NotImplementedError("dcn_head=True (deformable-conv head) is not part of the 3D-Dual-Fusion configs")
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.code_weights = code_weights
        self.weight = weight
        self.dataset = dataset
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.box_n_dim = 9 if 'vel' in common_heads else 7
        self.use_direction_classifier = False
        self.shared_conv = nn.Sequential(nn.Conv2d(in_channels, share_conv_channel, kernel_size=3, padding=1, bias=True),
                                         nn.BatchNorm2d(share_conv_channel), nn.ReLU(inplace=True))
        self.tasks = nn.ModuleList()
        for num_cls in num_classes:
            heads = copy.deepcopy(common_heads)
            heads.update(dict(hm=(num_cls, num_hm_conv)))
            self.tasks.append(SepHead(share_conv_channel, heads, bn=True, init_bias=init_bias, final_kernel=3))

    def forward(self, x, *kwargs):
        x = self.shared_conv(x)
        return [task(x) for task in self.tasks]

    def loss(self, example, preds_dicts, batch_dict=None, **kwargs):
        raise NotImplementedError("training rows are out of this build's scope (SURVEY.md section 8f row 4)")

    # ------------------------------------------------------------------ decode + NMS on the device
    @staticmethod
    def _rows(v):
        """NCHW head map -> [B*H*W, C] rows (the reference's own permute(0, 2, 3, 1), center_head.py:321-323); maps that
        are already channels-last views of rows are taken in place."""
        B, C, H, W = v.shape
        return v.permute(0, 2, 3, 1).reshape(B * H * W, C)

    @torch.no_grad()
    def predict_device(self, preds_dicts, test_cfg):
        """-> (boxes [T*B, post_max, 9|7], scores, labels (int32, class offsets applied), counts [T*B]) on the device;
        segment index = task * B + sample.  No host synchronisation."""
        get = test_cfg.get if hasattr(test_cfg, "get") else lambda k, d=None: getattr(test_cfg, k, d)
        if get('double_flip', False):
            raise NotImplementedError("double-flip test-time augmentation is not served by the device tail")
        if get('per_class_nms', False):
            raise NotImplementedError("per_class_nms is a no-op in the reference (center_head.py:431-432)")
        nms = get('nms')
        nms_get = nms.get if hasattr(nms, "get") else lambda k, d=None: getattr(nms, k, d)
        B, _, H, W = preds_dicts[0]['hm'].shape
        tasks, base = [], 0
        for t, pd in enumerate(preds_dicts):
            d = {k: self._rows(pd[k].float()) for k in ('hm', 'reg', 'height', 'dim', 'rot')}
            if 'vel' in pd:
                d['vel'] = self._rows(pd['vel'].float())
            d['label_base'] = base
            base += self.num_classes[t]
            tasks.append(d)
        rng = get('post_center_limit_range')
        rng = list(rng) if rng is not None and len(rng) > 0 else None
        if get('circular_nms', False):
            radii = list(get('min_radius'))
            if len(set(float(r) for r in radii)) != 1:
                # per-task radii: one call per task keeps the single-threshold ABI simple
                outs = [_ops.centerhead_predict([tk], B, H, W, get('out_size_factor'), get('voxel_size'), get('pc_range'),
                                                rng, get('score_threshold'), _ops.NMS_CIRCLE, float(radii[i]),
                                                min(4096, H * W), nms_get('nms_post_max_size'))
                        for i, tk in enumerate(tasks)]
                return tuple(torch.cat([o[j] for o in outs]) for j in range(4))
            mode, thr, pre = _ops.NMS_CIRCLE, float(radii[0]), min(4096, H * W)
        else:
            mode, thr, pre = _ops.NMS_ROTATED, float(nms_get('nms_iou_threshold')), int(nms_get('nms_pre_max_size'))
        return _ops.centerhead_predict(tasks, B, H, W, get('out_size_factor'), get('voxel_size'), get('pc_range'), rng,
                                       get('score_threshold'), mode, thr, pre, int(nms_get('nms_post_max_size')))

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """Reference return format: per sample {'box3d_lidar', 'scores', 'label_preds', 'metadata'} with the tasks'
        detections concatenated in task order (center_head.py:436-457).  One host round trip (the counts)."""
        boxes, scores, labels, counts = self.predict_device(preds_dicts, test_cfg)
        T = len(preds_dicts)
        B = boxes.shape[0] // T
        cnt = counts.view(T, B).cpu().tolist()
        metas = example.get("metadata", None) if isinstance(example, dict) else None
        if not metas:
            metas = [None] * B
        out = []
        for b in range(B):
            segs = [(t * B + b, cnt[t][b]) for t in range(T)]
            out.append({'box3d_lidar': torch.cat([boxes[s, :n] for s, n in segs]),
                        'scores': torch.cat([scores[s, :n] for s, n in segs]),
                        'label_preds': torch.cat([labels[s, :n] for s, n in segs]).long(),
                        'metadata': metas[b]})
        return out
