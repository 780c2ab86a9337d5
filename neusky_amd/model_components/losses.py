"""Closed-form losses of the train step (torch ops on small tensors): SURVEY.md section 8 A12."""
from __future__ import annotations

from typing import Optional, Dict, Sequence

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


def interlevel_per_ray(weights_list: Sequence[torch.Tensor], sbins_list: Sequence[torch.Tensor]):
    """per-ray interlevel sums of every proposal level ([R] each): interlevel_loss = sum of all of them / numel(final weights)"""
    from .. import ops
    c, w = sbins_list[-1].detach(), weights_list[-1].detach()
    return [ops.InterlevelFn.apply(c, w, sb.detach(), wp) for sb, wp in zip(sbins_list[:-1], weights_list[:-1])]


def interlevel_loss(weights_list: Sequence[torch.Tensor], sbins_list: Sequence[torch.Tensor]) -> torch.Tensor:
    """nerfstudio interlevel_loss (called neusky_model.py:987-988); weights [R,n], spacing bins [R,n+1]."""
    # one launch per proposal level each way (ops.InterlevelFn)
    per_ray = interlevel_per_ray(weights_list, sbins_list)
    return (per_ray[0] if len(per_ray) == 1 else torch.cat(per_ray)).sum() * (1.0 / weights_list[-1].numel())


class LossDict(dict):
    """scaled loss terms (differentiable entries: a trainer that sums the values, as nerfstudio's does, gets the same objective).
    `.parts` = [(x, coef | None, scale)]: the UNSCALED tensors the entries come from; total_loss() forms the objective from them in
    ONE launch each way (ops.TotalLossFn) instead of through the entries' multiplies, sums and adds (a dozen scalar launches each
    way).  `.total`: the objective once formed (or set by a caller)."""
    total: Optional[torch.Tensor] = None
    parts: Optional[list] = None


_COEF_CACHE: Dict[tuple, torch.Tensor] = {}


def total_loss(loss_dict: Dict[str, torch.Tensor]) -> torch.Tensor:
    t = getattr(loss_dict, "total", None)
    if t is not None:
        return t
    parts = getattr(loss_dict, "parts", None)
    if parts:
        from .. import ops
        total = None
        for i in range(0, len(parts), 8):  # (one launch takes up to eight pieces: both models' terms and the interlevel sums are five)
            metas, tensors = [], []
            for x, c, sc in parts[i:i + 8]:
                metas.append((c is not None, float(sc)))
                tensors.append(x)
                if c is not None:
                    tensors.append(c)
            t = ops.TotalLossFn.apply(tuple(metas), *tensors)
            total = t if total is None else total + t
        loss_dict.total = total
        return total
    return sum(loss_dict.values())


def merge_loss_dicts(a: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = LossDict({**a, **b})
    if not (set(a) & set(b)):
        pa, pb = getattr(a, "parts", None), getattr(b, "parts", None)
        if pa and pb and getattr(a, "total", None) is None and getattr(b, "total", None) is None:
            out.parts = list(pa) + list(pb)  # the objective of both from one launch (total_loss)
        else:
            out.total = total_loss(a) + total_loss(b)
    return out


def scale_dict(d: Dict[str, torch.Tensor], coefficients: Dict[str, float]) -> Dict[str, torch.Tensor]:
    """nerfstudio misc.scale_dict: only keys present in `coefficients` are scaled (so the reference's
    'eikonal_loss' entry never meets its 'eikonal loss' coefficient - reproduced, see oracle)."""
    keys = list(d)
    vals = [d[k] for k in keys]
    if len(keys) < 2 or not all(isinstance(v, torch.Tensor) and v.numel() == 1 for v in vals):
        return {k: (v * coefficients[k] if k in coefficients else v) for k, v in d.items()}
    dev = vals[0].device
    ck = (tuple(keys), tuple(float(coefficients.get(k, 1.0)) for k in keys), str(dev))
    coef = _COEF_CACHE.get(ck)
    if coef is None:
        coef = _COEF_CACHE[ck] = torch.tensor(ck[1], dtype=vals[0].dtype).to(dev)
    scaled = torch.stack([v.reshape(()) for v in vals]) * coef
    out = LossDict({k: scaled[i] for i, k in enumerate(keys)})
    out.total = scaled.sum()
    return out
