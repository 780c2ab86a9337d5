"""GPU parity of the render-stage kernels: golden vectors of the reference (G3/G4/G5) and the CPU oracle."""
import os

import numpy as np
import pytest
import torch

import inputs as gi
from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(__file__), "golden")
T = torch.from_numpy


def load(name):
    return dict(np.load(os.path.join(G, name + ".npz")))


def _hemi_gpu(inp, want_lin=False):
    from neusky_amd import hip
    R, S, _ = inp["albedo"].shape
    dv = lambda k: T(np.ascontiguousarray(inp[k])).to(DEV)
    rgb = torch.empty(R, 3, device=DEV)
    lin = torch.empty(R, 3, device=DEV)
    hip.hemi_composite_fwd(dv("albedo"), dv("normals"), dv("weights"), dv("dirs"), dv("cam_colours"),
                           T(inp["cam_of_ray"]).to(torch.int32).to(DEV), dv("vis"), dv("bg"), rgb, lin)
    return (rgb, lin) if want_lin else rgb


def test_hemi_composite_golden_small():
    g = load("g3_lambertian_small")
    rgb = _hemi_gpu(g).cpu().numpy()
    # tolerance from the north star: 1e-4 relative on rendered radiance
    np.testing.assert_allclose(rgb, g["rgb_train"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rgb, g["rgb_eval"], rtol=1e-4, atol=1e-6)


def test_hemi_composite_golden_big():
    g = load("g3_lambertian_big")
    R, S, D, U = [int(v) for v in g["shape"]]
    inp = gi.lambertian_inputs(seed=int(g["seed"]), R=R, S=S, D=D, U=U)
    rgb = _hemi_gpu(inp).cpu().numpy()
    np.testing.assert_allclose(rgb, g["rgb_train"], rtol=1e-4, atol=2e-6)


def test_hemi_composite_full_size_properties():
    """BASELINE size (1024 x 96 x 512): linearity in the light colours + oracle spot check."""
    from neusky_amd import hip
    inp = gi.lambertian_inputs(seed=5, R=1024, S=96, D=512, U=300)
    rgb, lin = _hemi_gpu(inp, want_lin=True)
    inp2 = dict(inp); inp2["cam_colours"] = inp["cam_colours"] * 2.0; inp2["bg"] = inp["bg"] * 2.0
    _, lin2 = _hemi_gpu(inp2, want_lin=True)
    assert torch.allclose(lin2, 2.0 * lin, rtol=1e-5, atol=1e-6)  # linear composite is linear in radiance
    sub = {k: (v[:32] if v.shape[0] == 1024 else v) for k, v in inp.items()}
    ref = O.lambertian_render(*(T(sub[k]).double() if sub[k].dtype != np.int64 else T(sub[k]) for k in
                                ["albedo", "normals", "dirs", "cam_colours", "cam_of_ray", "vis", "bg", "weights"]))
    np.testing.assert_allclose(rgb[:32].cpu().numpy(), ref.numpy(), rtol=1e-4, atol=2e-6)


def test_hemi_composite_backward_vs_autograd():
    from neusky_amd import hip
    inp = gi.lambertian_inputs(seed=8, R=24, S=10, D=100, U=5)
    inp["normals"][0, 0] = inp["normals"][0, 1]  # no exactly-zero normal (kink of the clamp)
    inp["bg"] *= 0.05; inp["cam_colours"] *= 0.05  # keep the composite inside the sRGB curve (< 1)
    keys = ["albedo", "normals", "cam_colours", "vis", "bg", "weights"]
    td = {k: T(inp[k]).double().requires_grad_(True) for k in keys}
    ref = O.lambertian_render(td["albedo"], td["normals"], T(inp["dirs"]).double(), td["cam_colours"], T(inp["cam_of_ray"]),
                              td["vis"], td["bg"], td["weights"])
    gr = torch.randn(ref.shape, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    grads = torch.autograd.grad((ref * gr).sum(), [td[k] for k in keys])
    dv = lambda k: T(np.ascontiguousarray(inp[k])).to(DEV)
    R, S, D, U = 24, 10, 100, 5
    rgb, lin = torch.empty(R, 3, device=DEV), torch.empty(R, 3, device=DEV)
    cam = T(inp["cam_of_ray"]).to(torch.int32).to(DEV)
    hip.hemi_composite_fwd(dv("albedo"), dv("normals"), dv("weights"), dv("dirs"), dv("cam_colours"), cam, dv("vis"), dv("bg"), rgb, lin)
    out = {"albedo": torch.empty(R, S, 3, device=DEV), "normals": torch.empty(R, S, 3, device=DEV),
           "weights": torch.empty(R, S, device=DEV), "cam_colours": torch.zeros(U, D, 3, device=DEV),
           "vis": torch.empty(R, D, device=DEV), "bg": torch.empty(R, 3, device=DEV)}
    hip.hemi_composite_bwd(dv("albedo"), dv("normals"), dv("weights"), dv("dirs"), dv("cam_colours"), cam, dv("vis"), dv("bg"), lin,
                           gr.float().to(DEV), out["albedo"], out["normals"], out["weights"], out["cam_colours"], out["vis"], out["bg"])
    for k, gref in zip(keys, grads):
        got = out[k].cpu().double()
        scale = gref.abs().max().item() + 1e-12
        assert (got - gref).abs().max().item() < 2e-4 * scale, (k, (got - gref).abs().max().item(), scale)


def _alpha_inputs(R, S, seed):
    g = torch.Generator().manual_seed(seed)
    sdf = torch.randn(R, S, generator=g) * 0.05
    grad = torch.nn.functional.normalize(torch.randn(R, S, 3, generator=g), dim=-1) * (1 + 0.1 * torch.randn(R, S, 1, generator=g))
    rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    bins = torch.sort(torch.rand(R, S + 1, generator=g) * 2 + 0.05, dim=1).values
    return sdf, grad, rd, bins[:, :-1].contiguous(), bins[:, 1:].contiguous(), torch.tensor([0.32])


@pytest.mark.parametrize("R,S", [(7, 5), (64, 96), (1024, 96), (33, 200)])
def test_neus_weights_fwd_bwd(R, S):
    from neusky_amd import hip
    sdf, grad, rd, st, en, var = _alpha_inputs(R, S, R + S)
    d = lambda t: t.to(DEV).contiguous()
    alpha, w = torch.empty(R, S, device=DEV), torch.empty(R, S, device=DEV)
    tb, acc, dep = torch.empty(R, device=DEV), torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    hip.neus_weights_fwd(d(sdf), d(grad), d(rd), d(st), d(en), d(var), 1.0, alpha, w, tb, acc, dep)
    sd, gd, vd = sdf.double().requires_grad_(True), grad.double().requires_grad_(True), var.double().requires_grad_(True)
    a_ref = O.neus_alpha(sd[..., None], gd, rd.double()[:, None, :].expand(R, S, 3), (en - st).double()[..., None], vd)
    w_ref, T_ref = O.weights_from_alphas(a_ref)
    assert (alpha.cpu().double() - a_ref[..., 0]).abs().max().item() < 2e-5
    assert (w.cpu().double() - w_ref[..., 0]).abs().max().item() < 2e-5
    assert (tb.cpu().double() - T_ref[:, -1, 0]).abs().max().item() < 2e-5
    mid = ((st + en) / 2).double()
    dref = (w_ref[..., 0] * mid).sum(1) / (w_ref[..., 0].sum(1) + 1e-10)
    assert (dep.cpu().double() - dref).abs().max().item() < 5e-5
    assert (acc.cpu().double() - w_ref[..., 0].sum(1)).abs().max().item() < 2e-5
    # backward
    g = torch.Generator().manual_seed(9)
    gw, gT = torch.randn(R, S, generator=g), torch.randn(R, generator=g)
    loss = (w_ref[..., 0] * gw.double()).sum() + (T_ref[:, -1, 0] * gT.double()).sum()
    gs_ref, gg_ref, gv_ref = torch.autograd.grad(loss, [sd, gd, vd])
    dsdf, dgrad, dvar = torch.empty(R, S, device=DEV), torch.empty(R, S, 3, device=DEV), torch.zeros(1, device=DEV)
    hip.neus_weights_bwd(d(sdf), d(grad), d(rd), d(st), d(en), d(var), 1.0, d(gw), d(gT), dsdf, dgrad, dvar)
    for got, ref in [(dsdf, gs_ref), (dgrad, gg_ref), (dvar, gv_ref)]:
        sc = ref.abs().max().item() + 1e-12
        assert (got.cpu().double() - ref).abs().max().item() < 5e-4 * sc, ((got.cpu().double() - ref).abs().max().item(), sc)


def test_visibility_geometry_golden():
    """nsky_visibility_rays + finish against the reference's compute_visibility (G4) and local frame (G5)."""
    from neusky_amd import hip
    for tag, lower in [("upper_lower1", 1.0), ("upper_lower0", 0.0)]:
        g = load(f"g4_visibility_{tag}")
        inp = gi.visibility_inputs(seed=4, R=16, S=3, D=42, n_outside=2)
        R, S, D = 16, 3, 42
        dirs = T(inp["dirs"])
        sel = torch.nonzero(dirs[:, 2] > 0)[:, 0]
        Dv = sel.numel()
        M = R * Dv
        sp = torch.empty(M, 3, device=DEV); xrow = torch.full((M, 16), float("nan"), device=DEV)
        sdist = torch.empty(M, device=DEV); tdist = torch.empty(M, device=DEV)
        hip.visibility_rays(T(inp["origins"][:, 0].copy()).to(DEV), T(inp["directions"][:, 0].copy()).to(DEV),
                            T(inp["depth"][:, 0].copy()).to(DEV), dirs[sel].contiguous().to(DEV), 1.0, sp, xrow, sdist, tdist)
        np.testing.assert_allclose(tdist.cpu().numpy(), g["termination_dist"], rtol=2e-5, atol=2e-6)
        # stand-in DDF of the generator evaluated on OUR sphere points (world dir = -l)
        dd = (-dirs[sel])[None].expand(R, Dv, 3).reshape(-1, 3)
        spc = sp.cpu()
        oz, dz = spc[:, 2], dd[:, 2]
        t = torch.where(dz < -1e-6, (0.1 - oz) / dz.clamp(max=-1e-6), torch.full_like(oz, 2.0)).clamp(0.0, 2.0)
        np.testing.assert_allclose(t.numpy(), g["expected_termination_dist"], rtol=1e-4, atol=1e-5)
        vis = torch.full((R, D), lower, device=DEV)
        hip.visibility_finish_fwd(t.to(DEV), sdist, torch.tensor([0.35], device=DEV), 25.0, sel.to(torch.int32).to(DEV), R, Dv, D, vis)
        ref = g["visibility"].reshape(R, S, D)[:, 0]
        np.testing.assert_allclose(vis.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
        # local direction + encoding row against the oracle (pinned by G5)
        dl = O.ddf_local_direction(spc.double(), dd.double())
        row = torch.cat([dl, O.nerf_encoding(dl, 2, 0.0, 2.0, False)], -1)
        got = xrow.cpu().double()
        assert (got[:, :15] - row).abs().max().item() < 2e-5
        assert (got[:, 15] == 0).all()
        # finish backward
        gv = torch.randn(R, D, generator=torch.Generator().manual_seed(1))
        dth, dthr = torch.empty(M, device=DEV), torch.zeros(1, device=DEV)
        hip.visibility_finish_bwd(t.to(DEV), sdist, torch.tensor([0.35], device=DEV), 25.0, sel.to(torch.int32).to(DEV), R, Dv, D,
                                  gv.to(DEV), dth, dthr)
        td = t.double().requires_grad_(True); thr = torch.tensor(0.35, dtype=torch.float64, requires_grad=True)
        v = 1 - torch.sigmoid(25.0 * (sdist.cpu().double() - td - thr))
        gt, gth = torch.autograd.grad((v.view(R, Dv) * gv[:, sel].double()).sum(), [td, thr])
        assert (dth.cpu().double() - gt).abs().max().item() < 1e-4 * gt.abs().max().item()
        assert abs(dthr.item() - gth.item()) < 1e-3 * abs(gth.item())


def test_density_weights_matches_torch_formulation():
    """nsky_density_weights_fwd/bwd == trunc_exp density + nerfstudio RaySamples.get_weights (the torch formulation kept
    in ray_samplers.weights_from_density) incl. the gradient w.r.t. the raw density head, for the proposal sizes 256 / 96,
    a ragged size, a padded head (ld 4) and saturated / overflowing densities (nan_to_num path)."""
    from neusky_amd import ops
    from neusky_amd.model_components.ray_samplers import weights_from_density
    dev = "cuda:0"
    torch.manual_seed(9)
    for R, n in ((64, 256), (37, 96), (5, 77)):
        raw = torch.randn(R * n, 4, device=dev) * 2.0
        raw[:, 1:] = 0.0
        raw[3, 0] = 40.0    # huge density: alpha = 1, later weights exactly 0
        raw[7, 0] = -30.0   # below trunc_exp's backward clamp (-15)
        ebins = torch.sort(torch.rand(R, n + 1, device=dev) * 2 + 0.05, dim=1).values
        probe = torch.randn(R, n, device=dev)
        r64 = raw.double().requires_grad_(True)
        dens = ops.TruncExpFn.apply(r64[:, :1]).view(R, n, 1)
        ref = weights_from_density(dens, (ebins[:, 1:] - ebins[:, :-1]).double()[..., None])[..., 0]
        (ref * probe.double()).sum().backward()
        r32 = raw.clone().requires_grad_(True)
        w = ops.DensityWeightsFn.apply(r32, ebins)
        assert w.shape == (R, n)
        assert (w.double() - ref).abs().max().item() < 2e-6
        (w * probe).sum().backward()
        assert (r32.grad[:, 1:] == 0).all()
        gref = r64.grad[:, 0]
        assert (r32.grad[:, 0].double() - gref).abs().max().item() < 1e-5 * max(gref.abs().max().item(), 1.0)


def test_ray_setup_kernels_match_oracle():
    """nsky_sphere_collider / nsky_uniform_bins / nsky_bins_to_samples against the oracle's SphereCollider and
    UniformSampler restatements: rays from inside and outside the unit sphere, rays that miss it, non-unit directions;
    jittered and plain lattices; mid-point positions."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import neusky_oracle as O
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(21)
    R = 777
    o = torch.randn(R, 3) * 0.8
    d = torch.randn(R, 3)
    d[: R // 2] = d[: R // 2] / d[: R // 2].norm(dim=1, keepdim=True)
    o[:40] = o[:40] * 6.0  # far outside: most of these miss the sphere
    nears, fars = hip.sphere_collider(o.to(dev), d.to(dev), 1.0, 0.05)
    rn, rf = O.sphere_collider(o.double(), d.double(), 1.0, 0.05)
    assert nears.shape == (R, 1) and fars.shape == (R, 1)
    assert (nears.cpu().double() - rn).abs().max().item() < 2e-5 and (fars.cpu().double() - rf).abs().max().item() < 2e-5
    assert bool((fars > nears).all())
    for n, jit in ((256, True), (96, False), (7, True)):
        j = torch.rand(R, 1) if jit else None
        sb, eb = hip.uniform_bins(nears, fars, n, None if j is None else j.reshape(-1).to(dev))
        rsb, reb = O.uniform_bins(nears.cpu().double(), fars.cpu().double(), n, None if j is None else j.double())
        assert sb.shape == (R, n + 1)
        assert (sb.cpu().double() - rsb).abs().max().item() < 3e-7
        assert (eb.cpu().double() - reb).abs().max().item() < 1e-5
        assert float(sb[:, 0].min()) >= 0.0 and float(sb[:, -1].max()) <= 1.0 and bool((sb[:, 1:] >= sb[:, :-1]).all())
        eb2, pos = hip.bins_to_samples(sb, nears, fars, o.to(dev), d.to(dev), want_ebins=True, want_positions=True)
        assert torch.equal(eb2, eb)
        mid = (reb[:, :-1] + reb[:, 1:]) / 2
        ref_pos = o.double()[:, None, :] + d.double()[:, None, :] * mid[..., None]
        assert (pos.cpu().double() - ref_pos).abs().max().item() < 1e-5 * max(1.0, ref_pos.abs().max().item())


def test_interlevel_kernel_matches_torch_formulation():
    """nsky_interlevel_fwd/bwd == nerfstudio interlevel_loss (the oracle's torch formulation, oracle.interlevel_loss / _outer) for the two proposal levels (256 and 96 bins against 96 final samples), value and gradient."""
    from neusky_amd import ops
    from neusky_amd.model_components import losses as L
    dev = "cuda:0"
    torch.manual_seed(13)
    R, S = 53, 96

    def bins(n):
        b = torch.sort(torch.rand(R, n + 1, device=dev), dim=1).values
        b[:, 0], b[:, -1] = 0.0, 1.0
        return b

    c = bins(S)
    w = torch.rand(R, S, device=dev) * 0.05
    w[:, 10:14] += 0.3
    for n in (256, 96, 5):
        sb = bins(n)
        wp = (torch.rand(R, n, device=dev) * 0.02)
        wp64 = wp.double().requires_grad_(True)
        w_outer = O._outer(c[..., :-1].double(), c[..., 1:].double(), sb[..., :-1].double(), sb[..., 1:].double(), wp64)
        ref = torch.mean(torch.clip(w.double() - w_outer, min=0) ** 2 / (w.double() + 1e-7))
        ref.backward()
        assert float(ref) > 0
        wp32 = wp.clone().requires_grad_(True)
        got = ops.InterlevelFn.apply(c, w, sb, wp32).sum() / w.numel()
        assert abs(float(got) - float(ref)) < 2e-5 * float(ref)
        got.backward()
        gref = wp64.grad
        assert (wp32.grad.double() - gref).abs().max().item() < 2e-5 * gref.abs().max().item()


def test_point_alphas_match_neus_weights_of_single_samples():
    """ops.PointAlphasFn (the hash-grid probe's three per-axis alphas) against ops.NeusWeightsFn on 3 P one-sample rays, values and
    every gradient (with a single sample the weight is the alpha)"""
    from neusky_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(21)
    P = 777
    sdf = (torch.randn(P, generator=g) * 0.05).to(dev)
    grad = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    gaps = [0.2, 0.15, 0.3]
    probe = torch.randn(P, 3, generator=g).to(dev)
    outs = []
    for which in (0, 1):
        s, gr = sdf.clone().requires_grad_(True), grad.clone().requires_grad_(True)
        var = torch.tensor([0.35], device=dev, requires_grad=True)
        if which == 0:
            a = ops.PointAlphasFn.apply(s, gr, dirs, gaps, var, 0.6)
        else:
            rep = lambda t, w: t.reshape(1, P, w).expand(3, P, w).reshape(3 * P, w)  # noqa: E731
            ends = torch.tensor(gaps, device=dev).reshape(3, 1, 1).expand(3, P, 1).reshape(3 * P, 1)
            w, _, _, _ = ops.NeusWeightsFn.apply(rep(s, 1), rep(gr, 3).reshape(3 * P, 1, 3), rep(dirs, 3), torch.zeros(3 * P, 1, device=dev),
                                                 ends, var, 0.6)
            a = w.reshape(3, P).t()
        (a * probe).sum().backward()
        outs.append((a.detach(), s.grad, gr.grad, var.grad))
    for x, y in zip(*outs):
        assert torch.allclose(x, y, rtol=1e-5, atol=2e-6 * float(y.abs().max())), float((x - y).abs().max())  # (the three gaps' terms sum in another order)
