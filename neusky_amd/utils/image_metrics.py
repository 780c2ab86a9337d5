"""Image metrics of the evaluation methods (neusky/models/neusky_model.py:1146-1154 uses torchmetrics' PeakSignalNoiseRatio(
data_range=1.0) and structural_similarity_index_measure; neither package nor their weights are part of the reference tree, so
the published definitions are restated here).  Inputs [1, C, H, W] in [0, 1]."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def psnr(target: torch.Tensor, pred: torch.Tensor, data_range: float = 1.0) -> torch.Tensor:
    mse = torch.mean((target - pred) ** 2)
    return 10.0 * torch.log10(data_range ** 2 / mse.clamp_min(1e-20))


def ssim(target: torch.Tensor, pred: torch.Tensor, data_range: float = 1.0, kernel_size: int = 11, sigma: float = 1.5,
         k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """mean SSIM with an 11x11 gaussian window (sigma 1.5), reflect padding, cropped to the valid interior"""
    C = target.shape[1]
    x = torch.arange(kernel_size, dtype=target.dtype, device=target.device) - (kernel_size - 1) / 2.0
    g = torch.exp(-(x / sigma) ** 2 / 2)
    g = (g / g.sum())[:, None] * (g / g.sum())[None, :]
    w = g.expand(C, 1, kernel_size, kernel_size).contiguous()
    pad = (kernel_size - 1) // 2
    if min(target.shape[-2:]) <= pad:
        return torch.tensor(float("nan"), device=target.device)
    a, b = F.pad(target, (pad,) * 4, mode="reflect"), F.pad(pred, (pad,) * 4, mode="reflect")
    mu_a, mu_b = F.conv2d(a, w, groups=C), F.conv2d(b, w, groups=C)
    s_aa = F.conv2d(a * a, w, groups=C) - mu_a ** 2
    s_bb = F.conv2d(b * b, w, groups=C) - mu_b ** 2
    s_ab = F.conv2d(a * b, w, groups=C) - mu_a * mu_b
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    s = ((2 * mu_a * mu_b + c1) * (2 * s_ab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (s_aa + s_bb + c2))
    return s[..., pad:-pad, pad:-pad].mean() if s.shape[-1] > 2 * pad and s.shape[-2] > 2 * pad else s.mean()


def grey_ramp(x: torch.Tensor) -> torch.Tensor:
    """[..., 1] scalar image in [0, 1] -> [..., 3] (stand-in for nerfstudio colormaps.apply_colormap)"""
    return x.clamp(0.0, 1.0).expand(*x.shape[:-1], 3)
