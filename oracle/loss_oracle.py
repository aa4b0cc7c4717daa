"""ORACLE (test infrastructure, not product code): the detection loss.

CPU fp32 restatement of loss/point_pillar_loss_multiclass.py: PointPillarLossMultiClass.forward :96-179, cls_loss_func :183-214,
sigmoid_cross_entropy_with_logits :251-276, add_sin_difference :279-293, WeightedSmoothL1Loss :13-76 (beta = 1/9, no code
weights).  Written with differentiable torch ops so that ``torch.autograd`` of this function is the checker of the
gradients av2x_pp_loss writes.  Parity: PINNED by tests/golden/loss_small.npz (tools/gen_golden.py runs the reference's own
loss class and its autograd).

As written in the reference: the classification sum is divided by the batch size inside cls_loss_func and again outside;
every non-positive anchor is a negative with weight 1 (``neg_equal_one`` is read but never used); the objectness term is a
plain mean BCE against ``pos_equal_one``.
"""
from __future__ import annotations

import torch


def smooth_l1(diff, beta=1.0 / 9.0):
    n = torch.abs(diff)
    return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)


def sigmoid_cross_entropy_with_logits(x, t):
    return torch.clamp(x, min=0) - x * t + torch.log1p(torch.exp(-torch.abs(x)))


def pp_loss(psm, rm, obj, targets, pos_equal_one, class_ids, num_class, cls_weight, reg_coe, alpha=0.25, gamma=2.0):
    """-> (total, reg, conf, obj) 0-dim tensors; inputs as PointPillarLossMultiClass.forward reads them."""
    B = psm.shape[0]
    cls_preds = psm.permute(0, 2, 3, 1).contiguous()
    obj_preds = obj.permute(0, 2, 3, 1).contiguous()
    labels = pos_equal_one.reshape(B, -1)
    positives, negatives = labels > 0, labels == 0
    cls_w = (negatives * 1.0 + 1.0 * positives).float()
    reg_w = positives.float()
    norm = torch.clamp(positives.sum(1, keepdim=True).float(), min=1.0)
    reg_w, cls_w = reg_w / norm, cls_w / norm
    onehot = torch.zeros(*class_ids.shape, num_class, dtype=cls_preds.dtype)
    onehot.scatter_(-1, class_ids.unsqueeze(-1).long(), 1.0)
    _, H, W, AC = cls_preds.shape
    A = AC // num_class
    x = cls_preds.view(B, H, W, A, num_class)
    t = onehot.view(B, H, W, A, num_class)
    w = cls_w.view(B, H, W, A, 1)
    p = torch.sigmoid(x)
    focal = (t * alpha + (1 - t) * (1 - alpha)) * torch.pow(t * (1.0 - p) + (1.0 - t) * p, gamma)
    cls_src = (focal * sigmoid_cross_entropy_with_logits(x, t) * w).sum() / B
    conf = cls_src.sum() / B * cls_weight
    r = rm.permute(0, 2, 3, 1).contiguous().view(B, -1, 7)
    tg = targets.view(B, -1, 7)
    pe = torch.sin(r[..., 6:7]) * torch.cos(tg[..., 6:7])
    te = torch.cos(r[..., 6:7]) * torch.sin(tg[..., 6:7])
    b1 = torch.cat([r[..., :6], pe], dim=-1)
    b2 = torch.cat([tg[..., :6], te], dim=-1)
    b2 = torch.where(torch.isnan(b2), b1, b2)
    reg = (smooth_l1(b1 - b2) * reg_w.unsqueeze(-1)).sum() / B * reg_coe
    s = torch.sigmoid(obj_preds)
    objl = (-(pos_equal_one * torch.log(s + 1e-6) + (1 - pos_equal_one) * torch.log(1 - s + 1e-6))).mean()
    return reg + conf + objl, reg, conf, objl
