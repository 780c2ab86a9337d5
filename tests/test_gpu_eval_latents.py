"""Eval-latent fitting loop on the device datamanager: the loss goes down, only the eval latents and
scale move, and the decoder / field / DDF parameters are untouched."""
import pytest
import torch

from util_step import randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fit_latent_codes_for_eval():
    from neusky_amd.data.image_datamanager import DeviceImageDataManager
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, images=4).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    m = pipe.model
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 16, 24
    images = torch.rand(N, H, W, 3, generator=g) * 0.5 + 0.25
    u = torch.rand(N, H, W, 4, generator=g)
    masks = torch.stack([u[..., 0] < 0.95, u[..., 1] < 0.5, u[..., 2] < 0.1, u[..., 1] >= 0.5], -1)  # sky = not fg
    c2w = torch.zeros(N, 3, 4)
    c2w[:, :, :3] = torch.eye(3)
    c2w[:, :, 3] = torch.tensor([[0.1, 0.0, 0.0], [-0.1, 0.1, 0.0]])
    dm = DeviceImageDataManager(images, masks, c2w, fx=30.0, fy=30.0, cx=W / 2, cy=H / 2, train_num_rays_per_batch=64, device=DEV, num_eval=2)
    before = {n: p.detach().clone() for n, p in pipe.named_parameters()}
    trace = m.fit_latent_codes_for_eval(dm, global_step=0, steps=60, log_every=1)
    losses = torch.stack(trace).cpu()
    assert torch.isfinite(losses).all()
    assert losses[-10:].mean() < losses[:5].mean(), (losses[:5], losses[-10:])
    moved = [n for n, p in pipe.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert sorted(moved) == ["_model.eval_illumination_latents", "_model.eval_scale"], moved
    assert not m.fitting_eval_latents and m.field.glin0.weight_v.requires_grad and m.field.encoding.params.requires_grad


def test_fit_matches_the_oracle_adam_steps():
    """N Adam steps of fit_latent_codes_for_eval on injected bundles and random draws against the float64
    restatement (oracle.neusky_eval_fit_loss + oracle.adam_fit): the fitted latents and scales agree"""
    from oracle import neusky_oracle as O
    from util_step import make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randoms_to
    torch.manual_seed(0)
    R, steps = 32, 6
    pipe = small_pipeline_config(R=R, num_prop=(24, 12), S=8, D=24, images=4).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    m = pipe.model
    dm = pipe.datamanager
    bundles = [dm.get_eval_image_half_bundle("full_image", image_index=i % 2, num_rays=R) for i in range(steps)]
    rnds = [make_randoms(pipe, R, seed=10 + i) for i in range(steps)]
    dev_rnds = []
    for r in rnds:
        d = randoms_to(r, DEV)
        for k in ("light_rotation", "grid_perturb", "grid_dirs"):
            d[k] = d[k].to(DEV)
        dev_rnds.append(d)
    m.set_step(10_000)
    trace = m.fit_latent_codes_for_eval(dm, global_step=10_000, steps=steps, bundles=bundles, randoms_per_step=dev_rnds, log_every=1)
    got_lat, got_scale = m.eval_illumination_latents.detach().cpu().double(), m.eval_scale.detach().cpu().double()
    # ---- oracle
    p = oracle_params(pipe)
    cfg = oracle_step_cfg(pipe)
    lat = torch.zeros_like(got_lat).requires_grad_(True)
    sc = torch.ones_like(got_scale).requires_grad_(True)

    def loss_fn(it):
        rb, batch = bundles[it]
        light = m.illumination_sampler(rotation=rnds[it]["light_rotation"]).double()
        return O.neusky_eval_fit_loss(p, cfg, rb.origins.cpu().double(), rb.directions.cpu().double(), rb.camera_indices.cpu().reshape(-1),
                                      batch["image"].cpu().double(), batch["mask"].cpu(), oracle_randoms(rnds[it], light), light, lat, sc)

    ref_trace = O.adam_fit([lat, sc], loss_fn, steps, lr=1e-1, lr_final=1e-7)
    got_trace = [float(t) for t in trace]
    for a, b in zip(got_trace, ref_trace):
        assert abs(a - b) < 2e-4 * max(abs(b), 1e-3), (got_trace, ref_trace)
    used = sorted({int(b[0].camera_indices.reshape(-1)[0]) for b in bundles})
    assert (got_lat[used] - lat.detach()[used]).abs().max() < 2e-3 * max(lat.detach()[used].abs().max().item(), 1e-3)
    assert (got_scale[used] - sc.detach()[used]).abs().max() < 2e-3


def test_fit_graph_replay_equals_eager():
    """the captured fitting iteration (one HIP graph replay per step + the Adam launches) reproduces the eager loop on the same
    injected bundles and random draws: loss trace and fitted latents agree to fp32 reduction-order noise"""
    from util_step import make_randoms, randoms_to
    torch.manual_seed(0)
    R, steps = 64, 12
    pipe = small_pipeline_config(R=R, num_prop=(32, 16), S=12, D=32, images=4).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    m = pipe.model
    bundles = [pipe.datamanager.get_eval_image_half_bundle("full_image", image_index=i % 2, num_rays=R) for i in range(3)]
    rnds = []
    for i in range(steps):
        d = randoms_to(make_randoms(pipe, R, seed=20 + i), DEV)
        for k in ("light_rotation", "grid_perturb", "grid_dirs"):
            d[k] = d[k].to(DEV)
        rnds.append(d)
    t_graph = m.fit_latent_codes_for_eval(pipe.datamanager, 10_000, steps=steps, bundles=bundles, randoms_per_step=rnds, log_every=1, use_graph=True)
    lat_graph, sc_graph = m.eval_illumination_latents.detach().clone(), m.eval_scale.detach().clone()
    t_eager = m.fit_latent_codes_for_eval(pipe.datamanager, 10_000, steps=steps, bundles=bundles, randoms_per_step=rnds, log_every=1, use_graph=False)
    g, e = torch.stack(t_graph).cpu(), torch.stack(t_eager).cpu()
    assert torch.isfinite(g).all() and float((g - e).abs().max()) < 1e-4 * float(e.abs().max()), (g, e)
    assert float((lat_graph - m.eval_illumination_latents.detach()).abs().max()) < 5e-3
    assert float((sc_graph - m.eval_scale.detach()).abs().max()) < 5e-3
    # and without injected draws the default path IS the graph path, and the loss goes down
    t = m.fit_latent_codes_for_eval(pipe.datamanager, 10_000, steps=40, bundles=bundles, log_every=1)
    t = torch.stack(t).cpu()
    assert torch.isfinite(t).all() and t[-5:].mean() < t[:3].mean()


def test_fit_after_a_train_step_uses_the_current_frozen_weights():
    """a train step leaves padded DDF weights cached from BEFORE the Adam update (which writes through raw
    pointers, so no version counter moves); the fit then freezes the DDF.  The cache must not survive that freeze: the weights the
    fit evaluates visibility with are the current ones, and no gradient work reaches the frozen DDF parameters."""
    from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
    from util_step import make_randoms, randoms_to
    torch.manual_seed(0)
    R = 32
    pipe = small_pipeline_config(R=R, num_prop=(24, 12), S=8, D=24, images=4).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    m = pipe.model
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    for g in opt.groups:  # a visible update of every group
        g.opt.lr = 1e-2
        g.sched = None
    rb, batch = pipe.datamanager.next_train(0)
    train_iteration(pipe, opt, 10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(make_randoms(pipe, R), DEV))
    ddf = m.visibility_field.field.ddf
    stale = [t.detach().clone() for t in ddf.padded_weights()]  # the train step's cache: built before the update
    bundles = [pipe.datamanager.get_eval_image_half_bundle("full_image", image_index=0, num_rays=R)]
    grads_before = [None if p.grad is None else p.grad.detach().clone() for p in ddf.parameters()]
    seen = {}
    orig = type(ddf).padded_weights

    def spy(self):
        wb = orig(self)
        if self is ddf and not any(p.requires_grad for p in self.parameters()):
            seen["wb"] = [t.detach().clone() for t in wb]
            seen["rg"] = [t.requires_grad for t in wb]
        return wb

    type(ddf).padded_weights = spy
    try:
        m.fit_latent_codes_for_eval(pipe.datamanager, 10_000, steps=3, bundles=bundles, use_graph=False)
    finally:
        type(ddf).padded_weights = orig
    assert "wb" in seen, "the fit never evaluated the (frozen) DDF"
    with torch.no_grad():
        fresh = ddf._padded_weights_uncached()
    assert any(not torch.equal(a, b) for a, b in zip(stale, fresh)), "the train step did not move the DDF weights: test is vacuous"
    for a, b in zip(seen["wb"], fresh):
        assert torch.equal(a, b), "the fit ran on weights cached before the optimizer update"
    assert not any(seen["rg"]), "frozen DDF copies still carry requires_grad: the fit would run the DDF weight-gradient backward"
    for p, g0 in zip(ddf.parameters(), grads_before):
        assert p.requires_grad  # restored
        if g0 is not None:
            assert torch.equal(p.grad, g0), "a gradient reached the frozen DDF parameters during the fit"
