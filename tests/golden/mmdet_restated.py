"""Pieces of mmdetection 2.10.0 (the version TF/README.md:56 pins; the package itself is a pip dependency that is
not vendored under /root/reference and not installed here) which TransFusionHead.loss calls, restated from their
published definitions so that the reference's OWN loss / target-assignment code (transfusion_head.py:1048-1283,
hungarian_assigner.py:100-160) can be imported and run by make_golden.py.  GENERATOR-SIDE ONLY: the fixture holds
the reference module's outputs; nothing here is imported by the product package.

  mmdet/models/losses/utils.py            reduce_loss / weight_reduce_loss / weighted_loss
  mmdet/models/losses/focal_loss.py       FocalLoss (sigmoid form; mmcv's sigmoid_focal_loss op, restated as the
                                          formula its CUDA kernel evaluates: mmcv/ops/csrc/sigmoid_focal_loss_cuda_kernel.cuh)
  mmdet/models/losses/smooth_l1_loss.py   L1Loss
  mmdet/models/losses/gaussian_focal_loss.py  GaussianFocalLoss
  mmdet/core/bbox/match_costs/match_cost.py   FocalLossCost
  mmdet/core/bbox/assigners/assign_result.py  AssignResult (fields only)
  mmdet/core/bbox/samplers/pseudo_sampler.py, sampling_result.py   PseudoSampler / SamplingResult"""
import torch


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {'none': loss, 'mean': loss.mean(), 'sum': loss.sum()}[reduction]
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def sigmoid_focal_loss_elements(pred, target, gamma, alpha):
    """mmcv SigmoidFocalLoss forward: target [N] class index (== C: background row of zeros)."""
    p = pred.sigmoid()
    flt_min = torch.finfo(torch.float32).tiny
    onehot = torch.zeros_like(pred)
    valid = target < pred.shape[1]
    onehot[valid, target[valid]] = 1.0
    pos = -alpha * (1 - p).pow(gamma) * torch.log(p.clamp(min=flt_min))
    neg = -(1 - alpha) * p.pow(gamma) * torch.log((1 - p).clamp(min=flt_min))
    return onehot * pos + (1 - onehot) * neg


class FocalLoss(torch.nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super().__init__()
        assert use_sigmoid
        self.use_sigmoid, self.gamma, self.alpha, self.reduction, self.loss_weight = use_sigmoid, gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        loss = sigmoid_focal_loss_elements(pred.contiguous(), target, self.gamma, self.alpha)
        if weight is not None and weight.shape != loss.shape:
            weight = weight.view(-1, 1) if weight.size(0) == loss.size(0) else weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction, avg_factor)


class L1Loss(torch.nn.Module):
    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert pred.size() == target.size() and target.numel() > 0
        return self.loss_weight * weight_reduce_loss(torch.abs(pred - target), weight,
                                                     reduction_override or self.reduction, avg_factor)


class GaussianFocalLoss(torch.nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        eps = 1e-12
        pos_weights = target.eq(1)
        neg_weights = (1 - target).pow(self.gamma)
        pos_loss = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_weights
        neg_loss = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_weights
        return self.loss_weight * weight_reduce_loss(pos_loss + neg_loss, weight, reduction_override or self.reduction,
                                                     avg_factor)


class VarifocalLoss(torch.nn.Module):
    """Constructed by the head (loss_iou) but never called (transfusion_head.py:1273-1279 are comments)."""

    def __init__(self, **kw):
        super().__init__()


LOSSES = dict(FocalLoss=FocalLoss, L1Loss=L1Loss, GaussianFocalLoss=GaussianFocalLoss, VarifocalLoss=VarifocalLoss,
              CrossEntropyLoss=VarifocalLoss)


def build_loss(cfg):
    cfg = dict(cfg)
    return LOSSES[cfg.pop('type')](**cfg)


class FocalLossCost(object):
    def __init__(self, weight=1., alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * self.weight


class AssignResult(object):
    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels


class SamplingResult(object):
    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_bboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_bboxes.numel() == 0:
            self.pos_gt_bboxes = torch.empty_like(gt_bboxes).view(-1, 4)
        else:
            if len(gt_bboxes.shape) < 2:
                gt_bboxes = gt_bboxes.view(-1, 4)
            self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


class PseudoSampler(object):
    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        neg_inds = torch.nonzero(assign_result.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)
