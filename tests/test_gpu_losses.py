"""Fused loss kernels (nsky_main_losses_fwd/bwd, nsky_ddf_losses_fwd/bwd) against the ORACLE's loss restatements
(oracle.neusky_losses / oracle.ddf_losses, pinned by golden G7 against the reference's get_loss_dict): every term and every
input gradient, float64 autograd as the reference.  Tolerances: terms 2e-5 relative, gradients 2e-5 of the tensor's max.
The two DDF option combinations outside the `neusky` config (mask_to_circumference, inverse_depth_weight) have no oracle
restatement and are checked against the formulas of ddf_model.py:413-436 written out in the test."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _oracle_main_terms(rgb, image, mask, eik, weights, normal, hdr_bg, grid, sdf_term, thr, alpha, target):
    from oracle import neusky_oracle as O
    ld = O.neusky_losses({"rgb": rgb, "eik_grad": eik, "weights": weights[..., None], "normal": normal, "hdr_background_colours": hdr_bg,
                          "grid_density": grid, "sdf_at_termination": sdf_term}, image, mask, thr.reshape(()), target_min_bias=target, sky_alpha=alpha)
    order = ("rgb_l1_loss", "eikonal_loss", "fg_mask_loss", "hashgrid_density_loss", "ground_plane_loss", "sky_pixel_loss",
             "visibility_sigmoid_loss", "sdf_level_set_visibility_loss")
    return torch.stack([ld[k].reshape(()) for k in order])


@pytest.mark.parametrize("R,S", [(64, 8), (1024, 96)])
def test_main_loss_terms_and_gradients(R, S):
    from neusky_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(R + S)
    rnd = lambda *s: torch.rand(*s, generator=g)
    rgb, image = rnd(R, 3), rnd(R, 3)
    mask = (rnd(R, 4) > 0.5).float()
    mask[:, 1] = mask[:, 1] * (1 - mask[:, 3])  # fg and sky exclusive
    eik = torch.randn(R, S, 3, generator=g)
    eik[0, 0] = 0.0  # ||g|| = 0: zero gradient
    # per-ray weight sums well inside (0.5), above (1.5) or below (0) the BCE clip interval [1e-3, 1 - 1e-3]: at the interval's
    # ends the fp32 sum decides the pass-through mask and 1 / ((1 - s) s) amplifies its rounding, in torch as much as here
    scale = torch.tensor([0.5, 1.5, 0.0])[torch.arange(R) % 3]
    weights = rnd(R, S) * (2.0 / S) * scale[:, None]
    normal = torch.randn(R, 3, generator=g) * 0.7
    hdr = torch.exp(torch.randn(R, 3, generator=g) * 2 - 2)  # both sRGB branches and the clamp at 1
    grid = torch.randn(50, 3, generator=g)
    sdf = torch.randn(777, 1, generator=g) * 0.1
    thr = torch.tensor([1.7])
    alpha, target = 0.1, 0.1
    ins64 = [t.double().requires_grad_(True) for t in (rgb, eik, weights, normal, hdr, grid, sdf, thr)]
    ref = _oracle_main_terms(ins64[0], image.double(), mask.double(), *ins64[1:7], ins64[7], alpha, target)
    wts = torch.linspace(0.5, 1.5, 8).double()
    gref = torch.autograd.grad((ref * wts).sum(), ins64)
    insg = [t.to(dev).requires_grad_(True) for t in (rgb, eik, weights, normal, hdr, grid, sdf, thr)]
    out = ops.MainLossesFn.apply(insg[0], image.to(dev), mask.to(dev), insg[1], insg[2], insg[3], insg[4], insg[5], insg[6], insg[7], alpha, target)
    assert torch.allclose(out.cpu().double(), ref.detach(), rtol=2e-5, atol=1e-7), (out.cpu(), ref)
    ggpu = torch.autograd.grad((out * wts.float().to(dev)).sum(), insg)
    for a, b, name in zip(ggpu, gref, ("rgb", "eik", "weights", "normal", "hdr", "grid", "sdf", "thr")):
        assert a.shape == b.shape
        err = (a.cpu().double() - b).abs().max().item()
        assert err <= 2e-5 * max(b.abs().max().item(), 1e-12), (name, err, b.abs().max().item())
    # absent inputs leave their terms at zero and need no gradient buffers
    part = ops.MainLossesFn.apply(insg[0], image.to(dev), mask.to(dev), None, None, None, None, None, None, None, alpha, target)
    assert torch.allclose(part[0], out[0]) and float(part[1:].abs().sum()) == 0.0


@pytest.mark.parametrize("circ,inv", [(False, False), (True, False), (False, True)])
def test_ddf_loss_terms_and_gradients(circ, inv):
    from neusky_amd import ops
    dev = "cuda:0"
    g = torch.Generator().manual_seed(3)
    Mr, Mm, Ms, radius = 1024, 300, 256, 1.0
    expected = torch.rand(Mr, generator=g) * 2
    term = torch.rand(Mr, 1, generator=g) * 2
    mask = (torch.rand(Mr, 1, generator=g) > 0.3).float()
    dw = torch.rand(Mr, generator=g)
    sdf = torch.randn(Mr, 1, generator=g) * 0.1
    mv_e, mv_t = torch.rand(Mm, generator=g) * 2, torch.rand(Mm, 1, generator=g) * 2
    sky_e, sky_t = torch.rand(Ms, generator=g) * 2, torch.rand(Ms, generator=g) * 2

    def torch_terms(e, s, me, se, tm, mvt):
        if circ:
            ex = e.unsqueeze(1); gt = tm.clone(); gt[mask == 0] = radius * 2
        else:
            ex = e.unsqueeze(1) * mask.double(); gt = tm * mask.double()
        iw = 1.0 / (gt + 1e-6) if inv else 1.0
        t0 = torch.mean(torch.abs(ex - gt) * dw.double().unsqueeze(-1) * iw)
        t1 = F.mse_loss(s * mask.double(), torch.zeros_like(s) * mask.double())
        t2 = F.l1_loss(s * mask.double(), torch.zeros_like(s) * mask.double())
        t3 = torch.mean(F.relu(me - mvt) ** 2)  # [Mm] - [Mm,1] -> [Mm,Mm] (sic, ddf_model.py:478-483)
        t4 = F.l1_loss(se, sky_t.double())
        return torch.stack([t0, t1, t2, t3, t4])

    ins64 = [t.double().requires_grad_(True) for t in (expected, sdf, mv_e, sky_e, term, mv_t)]  # the targets carry gradients too
    ref = torch_terms(*ins64)
    if not circ and not inv:  # the `neusky` configuration: the oracle's restatement (pinned by G7) must say the same
        from oracle import neusky_oracle as O
        o = O.ddf_losses(ins64[0], ins64[4], mask.double(), dw.double(), ins64[1], ins64[2], ins64[5], ins64[3], sky_t.double())
        assert torch.allclose(torch.stack([o["depth_l1_loss"], o["sdf_l2_loss"], o["multi_view_loss"], o["sky_ray_loss"]]), ref[[0, 1, 3, 4]], rtol=1e-12)
    wts = torch.tensor([1.0, 0.7, 1.3, 2.0, 0.5]).double()
    gref = torch.autograd.grad((ref * wts).sum(), ins64)
    insg = [t.to(dev).requires_grad_(True) for t in (expected, sdf, mv_e, sky_e, term, mv_t)]
    flags = dict(want_depth=1, want_sdf_l2=1, want_sdf_l1=1, mask_to_circumference=int(circ), inverse_depth_weight=int(inv), radius=radius)
    out = ops.DDFLossesFn.apply(insg[0], insg[4], mask.to(dev), dw.to(dev), insg[1], insg[2], insg[5], insg[3], sky_t.to(dev), flags)
    assert torch.allclose(out.cpu().double(), ref.detach(), rtol=3e-5, atol=1e-7), (out.cpu(), ref)
    ggpu = torch.autograd.grad((out * wts.float().to(dev)).sum(), insg)
    for a, b, name in zip(ggpu, gref, ("expected", "sdf", "mv", "sky", "term", "mv_term")):
        assert a.shape == b.shape
        err = (a.cpu().double() - b).abs().max().item()
        assert err <= 3e-5 * max(b.abs().max().item(), 1e-12), (name, err)


def test_train_metrics_kernel_matches_torch():
    DEV = "cuda:0"
    """hip.train_metrics against the torch expressions of neusky_model.py:1064-1072 and ddf_model.py:381-405"""
    import torch.nn.functional as F
    from neusky_amd import hip
    g = torch.Generator().manual_seed(12)
    rgb, image = torch.rand(1024, 3, generator=g), torch.rand(1024, 3, generator=g)
    v = torch.tensor([0.3])
    out = hip.train_metrics(rgb.to(DEV), image.to(DEV), None, 1.0, v.to(DEV)).cpu()
    assert abs(float(out[0]) - float(-10.0 * torch.log10(F.mse_loss(rgb.double(), image.double())))) < 1e-4
    s_val = torch.exp(v * 10.0).clip(1e-6, 1e6)
    assert torch.allclose(out[1], s_val[0], rtol=1e-6) and torch.allclose(out[2], 1.0 / s_val[0], rtol=1e-6)
    M = 262144 + 77
    pred, gt = torch.rand(M, generator=g) * 2, torch.rand(M, 1, generator=g) * 2
    mask = (torch.rand(M, 1, generator=g) > 0.3).float()
    got = float(hip.train_metrics(pred.to(DEV), gt.to(DEV), mask.to(DEV), 4.0)[0])
    want = float(10 * torch.log10(4.0 / F.mse_loss(pred.double().unsqueeze(1) * mask.double(), gt.double() * mask.double())))
    assert abs(got - want) < 1e-4
    # clip of the variance
    assert float(hip.train_metrics(rgb.to(DEV), image.to(DEV), None, 1.0, torch.tensor([5.0], device=DEV))[1]) == 1e6


def test_eval_branch_matches_the_reference_golden(golden_dir):
    """the evaluation / eval-latent-fitting branch of get_loss_dict (neusky_model.py:1036-1059) through the fused loss kernel,
    against the G7 vectors produced by the REFERENCE's own get_loss_dict (tests/golden/make_golden.py: `evalloss_*`) and against
    the oracle's restatement with its float64 input gradients"""
    import os
    import numpy as np
    from neusky_amd import ops
    from oracle import neusky_oracle as O
    dev = "cuda:0"
    g = np.load(os.path.join(golden_dir, "g7_losses.npz"))
    T = lambda k: torch.from_numpy(g[k])  # noqa: E731
    rgb, hdr = T("rgb").to(dev).requires_grad_(True), T("hdr_bg").to(dev).requires_grad_(True)
    out = ops.MainLossesFn.apply(rgb, T("image").to(dev), T("mask").float().to(dev), None, None, None, hdr, None, None, None, 0.1, 0.1)
    assert abs(float(out[0]) - float(g["evalloss_rgb_l1_loss"])) < 2e-5 * abs(float(g["evalloss_rgb_l1_loss"]))
    assert abs(float(out[5]) - float(g["evalloss_sky_pixel_loss"])) < 2e-5 * abs(float(g["evalloss_sky_pixel_loss"]))
    assert float(out[[1, 2, 3, 4, 6, 7]].abs().sum()) == 0.0
    r64, h64 = T("rgb").double().requires_grad_(True), T("hdr_bg").double().requires_grad_(True)
    ld = O.neusky_eval_losses(r64, h64, T("image").double(), T("mask"))
    gref = torch.autograd.grad(ld["rgb_l1_loss"] + 1.3 * ld["sky_pixel_loss"], [r64, h64])
    ggpu = torch.autograd.grad(out[0] + 1.3 * out[5], [rgb, hdr])
    for a, b in zip(ggpu, gref):
        assert (a.cpu().double() - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_one_launch_objective_equals_the_sum_of_the_entries():
    DEV = "cuda:0"
    """losses.total_loss forms the objective from the UNSCALED pieces in one launch each way (ops.TotalLossFn / nsky_weighted_total); the
    dictionary entries stay differentiable for a trainer that sums the values (nerfstudio: functools.reduce(torch.add, loss_dict.values())).
    Both must be the same number and give the same gradients; and the weighted-total kernel itself against torch on ragged segments."""
    from neusky_amd import ops
    from neusky_amd.engine import Optimizers, neusky_optimizers
    from neusky_amd.model_components.losses import total_loss
    from util_step import make_randoms, randomise, randoms_to, small_pipeline_config
    # ---- the kernel: three segments, with and without coefficient vectors
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(n, generator=g).to(DEV).requires_grad_(True) for n in (8, 1024, 3001)]
    cs = [torch.randn(8, generator=g).to(DEV), None, torch.rand(3001, generator=g).to(DEV)]
    sc = [1.0, 0.25 / 1024, -3.0]
    metas = tuple((c is not None, s) for c, s in zip(cs, sc))
    tens = [t for x, c in zip(xs, cs) for t in ((x, c) if c is not None else (x,))]
    got = ops.TotalLossFn.apply(metas, *tens)
    want = sum(((x.double() * c.double()).sum() if c is not None else x.double().sum()) * s for x, c, s in zip(xs, cs, sc))
    assert abs(float(got) - float(want)) <= 1e-5 * max(abs(float(want)), 1.0)
    (got * 2.0).backward()
    for x, c, s in zip(xs, cs, sc):
        ref = (c if c is not None else torch.ones_like(x)) * (2.0 * s)
        assert torch.allclose(x.grad, ref, rtol=1e-6, atol=1e-7)
    # ---- the step: objective and slab gradients, one-launch total against the summed entries
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=64, num_prop=(24, 12), S=8, D=24).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    rb, batch = pipe.datamanager.next_train(0)
    rnd = randoms_to(make_randoms(pipe, 64), DEV)
    slabs = []
    for mode in ("total", "sum"):
        opt.zero_grad_all()
        _, ld, _ = pipe.get_train_loss_dict(1000, ray_bundle=rb, batch=batch, randoms=rnd)
        assert getattr(ld, "parts", None), "the merged dictionary lost its unscaled pieces"
        loss = total_loss(ld) if mode == "total" else sum(ld.values())
        loss.backward()
        opt.collect_grads()
        slabs.append((float(loss), opt.flat_g.detach().clone()))
    (la, ga), (lb, gb) = slabs
    assert abs(la - lb) <= 2e-6 * max(abs(lb), 1.0), (la, lb)
    assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()), float((ga - gb).abs().max())
