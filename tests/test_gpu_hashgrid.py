"""GPU parity of the hash-grid encode kernels against the CPU oracle (tcnn HashGrid restatement)."""
import numpy as np
import pytest
import torch

from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(smooth, n_levels=16, log2T=19, max_res=2048, seed=0, table_scale=0.5):
    from neusky_amd.encoding import HashGridGeometry
    geom = HashGridGeometry(n_levels=n_levels, log2_hashmap_size=log2T, max_res=max_res, smoothstep=smooth)
    cfg = O.HashGridCfg(n_levels=n_levels, log2_hashmap_size=log2T, max_res=max_res, smoothstep=smooth)
    assert geom.offsets == cfg.offsets and geom.resolutions == cfg.resolutions
    g = torch.Generator().manual_seed(seed)
    table = (torch.rand(geom.n_params, 2, generator=g) * 2 - 1) * table_scale
    return geom, cfg, table


@pytest.mark.parametrize("mode", [0, 1])
def test_indices_bit_exact(mode):
    from neusky_amd import hip
    geom, cfg, table = _setup(False)
    g = torch.Generator().manual_seed(1)
    P = 4099
    x = (torch.rand(P, 3, generator=g) * 2 - 1) * (1.0 if mode == 0 else 1.3)
    x[:3] = torch.tensor([[1.0, -1.0, 0.0], [0.0, 0.0, 0.0], [0.999999, 0.5, -0.25]])
    idx = hip.hash_indices(geom, table.to(DEV), x.to(DEV), mode).cpu().to(torch.int64) & 0xFFFFFFFF
    pos = x if mode == 0 else (O.scene_contraction(x) + 2.0) / 4.0
    ref, _ = O.hash_grid_indices(pos, cfg)
    # cell choice depends on fp32 rounding of (contract(x)+2)/4 * scale + 0.5: identical op order -> bit exact
    mism = (idx != ref).any(-1).any(-1)
    assert mism.sum().item() == 0, f"{mism.sum().item()} points with different corner rows"


@pytest.mark.parametrize("smooth,mode", [(True, 1), (False, 0)])
def test_forward_rows_and_tangents(smooth, mode):
    from neusky_amd import hip
    geom, cfg, table = _setup(smooth)
    g = torch.Generator().manual_seed(2)
    P = 1000
    x = (torch.rand(P, 3, generator=g) * 2 - 1) * (1.25 if mode == 1 else 1.0)
    include_x, pe = (True, 6) if mode == 1 else (True, 0)
    width = 3 + 6 * pe + 32
    ldy = (width + 3) // 4 * 4
    Y = torch.full((P, ldy), float("nan"), device=DEV)
    Tn = torch.full((3, P, ldy), float("nan"), device=DEV)
    hip.encode_fwd(geom, table.to(DEV), x.to(DEV), mode, include_x, pe, 5.0, Y, Tn)

    xd = x.double().requires_grad_(True)
    pos = xd if mode == 0 else (O.scene_contraction(xd) + 2.0) / 4.0
    feat = O.hash_grid_encode(pos, table.double(), cfg)
    parts = [xd] + ([O.nerf_encoding(xd, 6, 0.0, 5.0, False)] if pe else []) + [feat]
    row = torch.cat(parts, -1)
    got = Y.cpu().double()
    # fp32 evaluation of t = frac(pos*scale + 0.5) at scale ~2047 carries ~1e-4 of a cell (1 ulp of pos*scale),
    # so the finest levels differ from exact math by ~1e-4 * |table| (table_scale = 0.5 here; 1e-4 at tcnn init).
    err = (got[:, :width] - row.detach()).abs()
    assert err.max().item() < 3e-4 and err.mean().item() < 1e-5, (err.max().item(), err.mean().item())
    assert (got[:, width:] == 0).all()
    # Jacobian rows: d row / d x_k by autograd, column by column (small P subset)
    sub = slice(0, 40)
    jac = torch.zeros(3, 40, width, dtype=torch.float64)
    for c in range(width):
        gr = torch.autograd.grad(row[sub, c].sum(), xd, retain_graph=True)[0][sub]
        jac[:, :, c] = gr.T
    tn = Tn.cpu().double()[:, sub, :width]
    scale = jac.abs().max().item()
    assert (tn - jac).abs().max().item() < 5e-4 * scale, ((tn - jac).abs().max().item(), scale)


# P = 777: direct scatter of the fine levels; P = 33000: chunk-owner kernel (>= 32768 points; T = 2^15 -> two chunks per hashed level)
@pytest.mark.parametrize("smooth,mode,with_t,P,log2T", [(True, 1, True, 777, 14), (False, 0, False, 777, 14),
                                                        (True, 1, True, 33000, 15), (False, 0, False, 33000, 15)])
def test_backward_table_and_input(smooth, mode, with_t, P, log2T):
    from neusky_amd import hip
    geom, cfg, table = _setup(smooth, n_levels=8, log2T=log2T, max_res=256)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(P, 3, generator=g) * 2 - 1) * (1.2 if mode == 1 else 1.0)
    pe = 6 if mode == 1 else 0
    width = 3 + 6 * pe + 2 * geom.n_levels
    ldy = (width + 3) // 4 * 4
    dY = torch.zeros(P, ldy); dY[:, :width] = torch.randn(P, width, generator=g)
    dT = None
    if with_t:
        dT = torch.zeros(3, P, ldy); dT[:, :, :width] = torch.randn(3, P, width, generator=g)
    dtab = torch.zeros(geom.n_params, 2, device=DEV)
    dx = torch.full((P, 3), float("nan"), device=DEV)
    hip.encode_bwd(geom, table.to(DEV), x.to(DEV), mode, True, pe, 5.0, dY.to(DEV), None if dT is None else dT.to(DEV), dtab, dx)

    xd = x.double().requires_grad_(True)
    tb = table.double().requires_grad_(True)
    pos = xd if mode == 0 else (O.scene_contraction(xd) + 2.0) / 4.0
    feat = O.hash_grid_encode(pos, tb, cfg)
    parts = [xd] + ([O.nerf_encoding(xd, 6, 0.0, 5.0, False)] if pe else []) + [feat]
    row = torch.cat(parts, -1)
    loss = (row * dY[:, :width].double()).sum()
    gx_ref = torch.autograd.grad(loss, xd, retain_graph=True, create_graph=False)[0]
    if with_t:
        # tangent rows = d row / d x_k ; add <dT_k, d row/dx_k>
        for k in range(3):
            ek = torch.zeros(P, 3, dtype=torch.float64); ek[:, k] = 1.0
            # forward-mode via double-backward trick
            v = torch.ones_like(row, requires_grad=True)
            gg = torch.autograd.grad(row, xd, v, create_graph=True)[0]
            jvp = torch.autograd.grad(gg, v, ek, create_graph=True)[0]  # d row / d x_k
            loss = loss + (jvp * dT[k, :, :width].double()).sum()
    gt_ref = torch.autograd.grad(loss, tb)[0]
    scale = gt_ref.abs().max().item()
    assert (dtab.cpu().double() - gt_ref).abs().max().item() < 5e-4 * scale
    sx = gx_ref.abs().max().item()
    assert (dx.cpu().double() - gx_ref).abs().max().item() < 5e-4 * sx


@pytest.mark.parametrize("smooth,mode,with_t,P", [(False, 0, False, 262144 + 1312), (True, 1, True, 98304), (True, 1, False, 263168)])
def test_owner_kernel_equals_direct_scatter_at_full_size(smooth, mode, with_t, P):
    """the step's grids (L = 16, T = 2^19, 16 -> 2048): table gradient from the chunk-owner kernel (one call, P >= 32768) against
    the direct scatter (no workspace handed in), accumulated into a table that is not zero"""
    from neusky_amd import hip
    from neusky_amd.encoding import HashGridGeometry
    geom = HashGridGeometry(smoothstep=smooth)
    g = torch.Generator().manual_seed(11)
    table = ((torch.rand(geom.n_params, 2, generator=g) * 2 - 1) * 1e-2).to(DEV)
    x = torch.rand(P, 3, generator=g) * 2 - 1
    x = torch.nn.functional.normalize(x, dim=-1) if mode == 0 else x * 1.3  # sphere points (DDF) / contracted scene points
    x = x.to(DEV).contiguous()
    pe = 6 if mode == 1 else 0
    width = 3 + 6 * pe + 2 * geom.n_levels
    ldy = (width + 3) // 4 * 4
    dY = torch.randn(P, ldy, generator=g).to(DEV)
    dT = torch.randn(3, P, ldy, generator=g).to(DEV) if with_t else None
    base = torch.randn(geom.n_params, 2, generator=g).to(DEV) * 1e-3
    a = base.clone()
    hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, dT, a, None)
    b = base.clone()
    hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, dT, b, None, workspace=None)  # no scratch: the direct scatter
    torch.cuda.synchronize()
    scale = float((b - base).abs().max())
    assert scale > 0
    err = float((a - b).abs().max())
    assert err < 2e-5 * scale, (err, scale)
    # every level received gradient through both paths
    for lvl in range(geom.n_levels):
        sl = slice(geom.offsets[lvl], geom.offsets[lvl + 1])
        assert float((a[sl] - base[sl]).abs().max()) > 0


def test_empty_and_bad_args():
    from neusky_amd import hip
    geom, cfg, table = _setup(False, n_levels=4, log2T=12, max_res=64)
    x = torch.zeros(0, 3, device=DEV)
    Y = torch.zeros(0, 12, device=DEV)
    hip.encode_fwd(geom, table.to(DEV), x, 0, True, 0, 0.0, Y)  # P == 0 is a no-op
    with pytest.raises(hip.NeuSkyHipError):
        hip.encode_fwd(geom, table.to(DEV), torch.zeros(4, 3, device=DEV), 0, True, 0, 0.0, torch.zeros(4, 8, device=DEV))


def test_frozen_table_backward_forms_only_the_input_gradient():
    """a table that does not require grad (the eval-latent fit): no scatter, dx as with a trainable table"""
    from neusky_amd import ops
    from neusky_amd.encoding import HashGridGeometry
    geom = HashGridGeometry(smoothstep=True)
    torch.manual_seed(0)
    table = ((torch.rand(geom.n_params, 2) * 2 - 1) * 1e-2).to("cuda:0")
    x = (torch.rand(40000, 3) * 1.6 - 0.8).to("cuda:0")
    probe = torch.randn(40000, 76).to("cuda:0")
    outs = []
    for trainable in (True, False):
        t = table.clone().requires_grad_(trainable)
        xi = x.clone().requires_grad_(True)
        y = ops.HashEncodeFn.apply(xi, t, geom, 1, True, 6, 5.0, False, True)
        (y * probe[:, :y.shape[1]]).sum().backward()
        outs.append(xi.grad)
        assert (t.grad is not None) == trainable
    assert torch.equal(outs[0], outs[1])
