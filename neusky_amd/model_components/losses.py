"""Closed-form losses of the train step (torch ops on small tensors): SURVEY.md section 8 A12."""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F


class RENISkyPixelLoss:
    """neusky/model_components/losses.py:44-58"""

    def __init__(self, alpha: float = 1.0):
        self.alpha = alpha

    def __call__(self, inputs, targets, mask):
        inputs = inputs * mask
        targets = targets * mask
        mse = F.mse_loss(inputs, targets)
        similarity = F.cosine_similarity(inputs, targets, dim=1, eps=1e-20)
        return mse + self.alpha * (1 - similarity.mean())


def monosdf_normal_loss(normal_pred: torch.Tensor, normal_gt: torch.Tensor) -> torch.Tensor:
    """nerfstudio monosdf_normal_loss (called neusky_model.py:1000)"""
    normal_gt = F.normalize(normal_gt, p=2, dim=-1)
    normal_pred = F.normalize(normal_pred, p=2, dim=-1)
    return torch.abs(normal_pred - normal_gt).sum(dim=-1).mean() + (1.0 - (normal_pred * normal_gt).sum(-1)).mean()


def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.clamp(torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1, 0, y1.shape[-1] - 1)
    idx_hi = torch.clamp(torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right"), 0, y1.shape[-1] - 1)
    return torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)


def interlevel_loss(weights_list: Sequence[torch.Tensor], sbins_list: Sequence[torch.Tensor]) -> torch.Tensor:
    """nerfstudio interlevel_loss (called neusky_model.py:987-988); weights [R,n], spacing bins [R,n+1]."""
    c, w = sbins_list[-1].detach(), weights_list[-1].detach()
    loss = 0.0
    for sb, wp in zip(sbins_list[:-1], weights_list[:-1]):
        w_outer = _outer(c[..., :-1], c[..., 1:], sb[..., :-1], sb[..., 1:], wp)
        loss = loss + torch.mean(torch.clip(w - w_outer, min=0) ** 2 / (w + 1e-7))
    return loss


def scale_dict(d: Dict[str, torch.Tensor], coefficients: Dict[str, float]) -> Dict[str, torch.Tensor]:
    """nerfstudio misc.scale_dict: only keys present in `coefficients` are scaled (so the reference's
    'eikonal_loss' entry never meets its 'eikonal loss' coefficient - reproduced, see oracle)."""
    return {k: (v * coefficients[k] if k in coefficients else v) for k, v in d.items()}
