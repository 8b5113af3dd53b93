"""Target assignment and loss functions of `TransFusionHead.loss`.

From the reference tree: `HungarianAssigner3D`, `BBoxBEVL1Cost`, `BBox3DL1Cost`, `IoU3DCost`
(TF/mmdet3d/core/bbox/assigners/hungarian_assigner.py:14-48,100-160), `gaussian_radius` / `draw_heatmap_gaussian`
(TF/mmdet3d/core/utils/gaussian.py:5-86), `clip_sigmoid` (TF/mmdet3d/models/utils/clip_sigmoid.py).
From mmdetection 2.10.0, which TF/README.md:56 pins and the reference imports but does not vendor (published
definitions): `FocalLoss` (sigmoid form), `L1Loss`, `GaussianFocalLoss` with `weight_reduce_loss`
(mmdet/models/losses/), `FocalLossCost` (mmdet/core/bbox/match_costs/match_cost.py), `AssignResult`, `PseudoSampler`.

These are the plain-torch (autograd) formulations; on the device the head's `loss` runs the same arithmetic in
csrc/tfloss.hip (`TransFusionHead.loss_device`).  The Hungarian matching itself is `scipy.optimize.
linear_sum_assignment` on the host, as in the reference (hungarian_assigner.py:133-138)."""
import numpy as np
import torch
from torch import nn

from .box3d import BboxOverlaps3D

try:
    from scipy.optimize import linear_sum_assignment
except ImportError:                                         # same behaviour as the reference: fail at assign()
    linear_sum_assignment = None


def clip_sigmoid(x, eps=1e-4):
    """In place on `x`, like the reference (the dense heat-map prediction holds probabilities afterwards)."""
    return torch.clamp(x.sigmoid_(), min=eps, max=1 - eps)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        if reduction == 'none':
            return loss
        return loss.mean() if reduction == 'mean' else loss.sum()
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class FocalLoss(nn.Module):
    """Sigmoid focal loss over class-index targets (index == num_classes: background)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super(FocalLoss, self).__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        p = pred.sigmoid()
        tiny = torch.finfo(torch.float32).tiny
        hit = target.view(-1, 1) == torch.arange(pred.shape[1], device=pred.device).view(1, -1)
        loss = torch.where(hit, -self.alpha * (1 - p).pow(self.gamma) * torch.log(p.clamp(min=tiny)),
                           -(1 - self.alpha) * p.pow(self.gamma) * torch.log((1 - p).clamp(min=tiny)))
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction, avg_factor)


class L1Loss(nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super(L1Loss, self).__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        assert pred.size() == target.size() and target.numel() > 0
        return self.loss_weight * weight_reduce_loss(torch.abs(pred - target), weight,
                                                     reduction_override or self.reduction, avg_factor)


class GaussianFocalLoss(nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super(GaussianFocalLoss, self).__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        eps = 1e-12
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * target.eq(1)
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * (1 - target).pow(self.gamma)
        return self.loss_weight * weight_reduce_loss(pos + neg, weight, reduction_override or self.reduction, avg_factor)


class _UnusedLoss(nn.Module):
    """`loss_iou` is constructed by the head and never evaluated (transfusion_head.py:1273-1279 are comments)."""

    def __init__(self, **kwargs):
        super(_UnusedLoss, self).__init__()
        self.loss_weight = kwargs.get('loss_weight', 1.0)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("loss_iou is not part of TransFusionHead.loss")


_LOSSES = dict(FocalLoss=FocalLoss, L1Loss=L1Loss, GaussianFocalLoss=GaussianFocalLoss, VarifocalLoss=_UnusedLoss,
               CrossEntropyLoss=_UnusedLoss)


def build_loss(cfg):
    if cfg is None:
        return None
    cfg = dict(cfg)
    kind = cfg.pop('type', None)
    if kind is None:                                        # bare dict(use_sigmoid=True): nothing to build
        return None
    if kind not in _LOSSES:
        raise KeyError("%s is not a loss of the TransFusion head" % kind)
    return _LOSSES[kind](**cfg)


# ------------------------------------------------------------------ matching costs
class FocalLossCost(object):
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        """cls_pred [num_query, num_class] logits, gt_labels [num_gt] -> [num_query, num_gt]."""
        p = cls_pred.sigmoid()
        neg = -(1 - p + self.eps).log() * (1 - self.alpha) * p.pow(self.gamma)
        pos = -(p + self.eps).log() * self.alpha * (1 - p).pow(self.gamma)
        return (pos[:, gt_labels] - neg[:, gt_labels]) * self.weight


class BBox3DL1Cost(object):
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        return torch.cdist(bboxes, gt_bboxes, p=1) * self.weight


class BBoxBEVL1Cost(object):
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, bboxes, gt_bboxes, train_cfg):
        rng = train_cfg['point_cloud_range']
        start = bboxes.new_tensor(rng[0:2])
        extent = bboxes.new_tensor(rng[3:5]) - bboxes.new_tensor(rng[0:2])
        return torch.cdist((bboxes[:, :2] - start) / extent, (gt_bboxes[:, :2] - start) / extent, p=1) * self.weight


class IoU3DCost(object):
    def __init__(self, weight):
        self.weight = weight

    def __call__(self, iou):
        return -iou * self.weight


_COSTS = dict(FocalLossCost=FocalLossCost, BBox3DL1Cost=BBox3DL1Cost, BBoxBEVL1Cost=BBoxBEVL1Cost, IoU3DCost=IoU3DCost)
_IOU_CALCULATORS = dict(BboxOverlaps3D=BboxOverlaps3D)


def _build(cfg, table):
    cfg = dict(cfg)
    return table[cfg.pop('type')](**cfg)


class AssignResult(object):
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult(object):
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :] if gt_bboxes.numel() else gt_bboxes.new_zeros((0, gt_bboxes.shape[-1]))


class PseudoSampler(object):
    """Every assigned query is a sample (mmdet pseudo_sampler.py)."""

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        return SamplingResult(pos, neg, bboxes, gt_bboxes, assign_result)


class HungarianAssigner3D(object):
    """One-to-one matching of proposals to ground truth on the summed classification / BEV-centre / 3-D IoU cost
    (hungarian_assigner.py:100-160)."""

    def __init__(self, cls_cost=dict(type='ClassificationCost', weight=1.), reg_cost=dict(type='BBoxBEVL1Cost', weight=1.0),
                 iou_cost=dict(type='IoU3DCost', weight=1.0), iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar')):
        if cls_cost.get('type') not in _COSTS:
            raise NotImplementedError("cls_cost %r: the 3D-Dual-Fusion configs use FocalLossCost" % (cls_cost.get('type'),))
        self.cls_cost, self.reg_cost, self.iou_cost = _build(cls_cost, _COSTS), _build(reg_cost, _COSTS), _build(iou_cost, _COSTS)
        self.iou_calculator = _build(iou_calculator, _IOU_CALCULATORS)

    def cost_matrix(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        iou = self.iou_calculator(bboxes, gt_bboxes)
        return (self.cls_cost(cls_pred[0].T, gt_labels) + self.reg_cost(bboxes, gt_bboxes, train_cfg) + self.iou_cost(iou)), iou

    @staticmethod
    def match(cost):
        """cost [num_query, num_gt] on the host -> (query indices, gt indices)."""
        if linear_sum_assignment is None:
            raise ImportError('Please run "pip install scipy" to install scipy first.')
        return linear_sum_assignment(cost)

    def assign(self, bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg):
        num_gts, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
        gt_inds = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)
        labels = bboxes.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            if num_gts == 0:
                gt_inds[:] = 0
            return AssignResult(num_gts, gt_inds, None, labels=labels)
        cost, iou = self.cost_matrix(bboxes, gt_bboxes, gt_labels, cls_pred, train_cfg)
        rows, cols = self.match(cost.detach().cpu())
        rows = torch.from_numpy(np.asarray(rows)).to(bboxes.device)
        cols = torch.from_numpy(np.asarray(cols)).to(bboxes.device)
        gt_inds[:] = 0
        gt_inds[rows] = cols + 1
        labels[rows] = gt_labels[cols]
        max_overlaps = torch.zeros_like(iou.max(1).values)
        max_overlaps[rows] = iou[rows, cols]
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)


def build_assigner(cfg):
    cfg = dict(cfg)
    kind = cfg.pop('type')
    if kind != 'HungarianAssigner3D':
        raise NotImplementedError("assigner %r: the 3D-Dual-Fusion configs use HungarianAssigner3D" % (kind,))
    return HungarianAssigner3D(**cfg)


# ------------------------------------------------------------------ dense heat-map targets
def gaussian_radius(det_size, min_overlap=0.5):
    """CenterNet's box-size -> splat radius rule (gaussian.py:60-86), element-wise over tensors of box (length, width)
    in fp32: the smallest positive root of three quadratics, one per overlap case."""
    height, width = det_size
    total = height + width
    roots = []                                               # operand order as the reference writes it (fp32 rounding)
    for a, b, c in ((1, total, width * height * (1 - min_overlap) / (1 + min_overlap)),
                    (4, 2 * total, (1 - min_overlap) * width * height),
                    (4 * min_overlap, -2 * min_overlap * total, (min_overlap - 1) * width * height)):
        roots.append((b + torch.sqrt(b ** 2 - 4 * a * c)) / 2)
    return torch.minimum(torch.minimum(roots[0], roots[1]), roots[2])


def gaussian_patch(radius):
    """[2r+1, 2r+1] fp32 samples of exp(-(x^2 + y^2) / (2 sigma^2)), sigma = (2r+1)/6, evaluated in fp64 and flushed
    to zero below eps * peak (gaussian.py:5-22)."""
    t = torch.arange(-radius, radius + 1, dtype=torch.float64)
    sigma = (2 * radius + 1) / 6
    g = torch.exp(-(t[None, :] ** 2 + t[:, None] ** 2) / (2 * sigma * sigma))
    g[g < torch.finfo(torch.float64).eps * g.max()] = 0
    return g.float()


def draw_heatmap_gaussian(heatmap, center, radius, k=1):
    """heatmap [H, W] <- max(heatmap, k * patch centred on pixel (x, y) = center), clipped at the borders
    (gaussian.py:25-57).  Host loop form; the device path splats every box of a batch in one launch
    (`df3d_draw_heatmap_gaussian`)."""
    x, y = int(center[0]), int(center[1])
    H, W = heatmap.shape[0:2]
    x0, x1, y0, y1 = max(x - radius, 0), min(x + radius + 1, W), max(y - radius, 0), min(y + radius + 1, H)
    if x1 > x0 and y1 > y0:
        patch = gaussian_patch(radius)[y0 - y + radius:y1 - y + radius, x0 - x + radius:x1 - x + radius]
        region = heatmap[y0:y1, x0:x1]
        region.copy_(torch.maximum(region, patch.to(heatmap.device) * k))
    return heatmap
