"""BASELINE config 1: RENI-shaped env-map decode, latent 36 x 3, 64 x 128 equirectangular directions.
The CPU oracle is the torch-CPU leg; the HIP decoder must match it, and rotating latent + directions about z
together must leave the map unchanged (the SO(2) equivariance RENI++ is built on)."""
import math

import pytest
import torch

from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def equirect_directions(h=64, w=128):
    phi = (torch.arange(h, dtype=torch.float64) + 0.5) / h * math.pi  # polar
    theta = (torch.arange(w, dtype=torch.float64) + 0.5) / w * 2 * math.pi
    P, T_ = torch.meshgrid(phi, theta, indexing="ij")
    return torch.stack([torch.sin(P) * torch.cos(T_), torch.sin(P) * torch.sin(T_), torch.cos(P)], -1).reshape(-1, 3)


def _params(field):
    p = {}
    net = field.network
    lins = net.mapping_network.linears()
    c = lambda t: t.detach().cpu().double()
    for i, lin in enumerate(lins[:-1]):
        p[f"reni.map_w{i}"], p[f"reni.map_b{i}"] = c(lin.weight), c(lin.bias)
    p["reni.map_wo"], p["reni.map_bo"] = c(lins[-1].weight), c(lins[-1].bias)
    for i, l in enumerate(net.net):
        p[f"reni.film_w{i}"], p[f"reni.film_b{i}"] = c(l.layer.weight), c(l.layer.bias)
    p["reni.out_w"], p["reni.out_b"] = c(net.final_layer.weight), c(net.final_layer.bias)
    return p


def test_envmap_decode_matches_cpu_oracle_and_is_so2_invariant():
    from neusky_amd.model_components.illumination import RENIField, RENIFieldConfig
    torch.manual_seed(0)
    field = RENIField(RENIFieldConfig(latent_dim=36)).to(DEV)
    dirs = equirect_directions()  # 8192 directions
    Z = torch.randn(36, 3, dtype=torch.float64) * 0.4
    scale = torch.tensor(1.3, dtype=torch.float64)
    with torch.no_grad():
        got = field.forward_grid(dirs.float().to(DEV), Z.float().to(DEV)[None], scale.float().to(DEV)[None])[0].cpu().double()
    ref = O.reni_decode(Z[None].expand(dirs.shape[0], -1, -1), dirs, scale.expand(dirs.shape[0]), _params(field))
    rel = ((got - ref).abs().max() / ref.abs().max()).item()
    assert rel < 1e-4, rel
    a = 1.1
    Rz = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]], dtype=torch.float64)
    with torch.no_grad():
        rot = field.forward_grid((dirs @ Rz.T).float().to(DEV), (Z @ Rz.T).float().to(DEV)[None], scale.float().to(DEV)[None])[0].cpu().double()
    assert ((rot - got).abs().max() / got.abs().max()).item() < 1e-4
