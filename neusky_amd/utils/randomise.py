"""Move a freshly built pipeline's parameters away from their (partly degenerate) initial values, so that every path of
the step carries signal: hash tables at U(-1e-4, 1e-4) and zero latents would leave most gradients numerically dead.
Used by bench.py (synthetic "mid-training" state) and by the parity tests (tests/util_step.py re-exports it)."""
from __future__ import annotations

import torch


def randomise(pipeline, seed=0, scale=1.0):
    """move parameters away from their (partly degenerate) initial values so every path carries signal"""
    g = torch.Generator().manual_seed(seed)
    m = pipeline.model
    with torch.no_grad():
        for name, p in pipeline.named_parameters():
            if name.endswith("encoding.params") or name.endswith("position_encoding.params"):
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.02 * scale).to(p.device))
        m.train_illumination_latents.copy_((torch.randn(m.train_illumination_latents.shape, generator=g) * 0.3).to(m.device))
        m.train_scale.copy_((1 + 0.2 * torch.rand(m.train_scale.shape, generator=g)).to(m.device))
        m.visibility_threshold.fill_(0.3)
        # geo layer 0 sees only x at geometric init (PE / hash columns zero): perturb so they matter
        w = m.field.glin0.weight_v
        w.add_((torch.randn(w.shape, generator=g) * 0.02).to(w.device))
        for net in m.proposal_networks:
            net.lin1.bias.fill_(1.0)
