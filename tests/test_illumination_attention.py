"""RENI++ attention-conditioned illumination decoder (neusky_config.py:78-95 selects conditioning="Attention"): the
ns_reni source is absent from the reference, so this is parity-unpinned own-definition code; what is checked: the product's form
(tokens linear in the direction, per-camera K / V, batched library GEMMs) against the oracle's plain per-pair restatement in
float64 -- values and latent gradients --, SO(2) equivariance about z, and that the reference's configuration constructs and trains."""
import math

import pytest
import torch

from oracle import neusky_oracle as O


def attn_params(net, dtype=torch.float64):
    c = lambda t: t.detach().cpu().to(dtype)  # noqa: E731
    p = {"reni.attn.token_w": c(net.token_embed.weight), "reni.attn.token_b": c(net.token_embed.bias),
         "reni.attn.query_w": c(net.query_embed.weight), "reni.attn.query_b": c(net.query_embed.bias),
         "reni.attn.lnf_w": c(net.ln_f.weight), "reni.attn.lnf_b": c(net.ln_f.bias), "reni.attn.out_w": c(net.out.weight), "reni.attn.out_b": c(net.out.bias)}
    for l, blk in enumerate(net.layers):
        for name in ("ln1", "ln2", "wq", "wk", "wv", "wo", "ff1", "ff2"):
            m = getattr(blk, name)
            p[f"reni.attn.l{l}.{name}_w"], p[f"reni.attn.l{l}.{name}_b"] = c(m.weight), c(m.bias)
    return p


def _field(L=12, device="cpu", seed=0):
    from neusky_amd.model_components.illumination import RENIFieldConfig
    torch.manual_seed(seed)
    f = RENIFieldConfig(conditioning="Attention", latent_dim=L).setup().to(device)
    with torch.no_grad():  # biases / norms away from their trivial init so every term is exercised
        for n_, p_ in f.network.named_parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1)
    return f


def _check(device, D=9):
    L, U = 12, 4
    f = _field(L, device)
    g = torch.Generator().manual_seed(1)
    lat = (torch.randn(U, L, 3, generator=g) * 0.6).to(device).requires_grad_(True)
    dirs = torch.randn(D, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=-1, keepdim=True)).to(device)
    sc = (torch.rand(U, generator=g) + 0.5).to(device)
    got = f.forward_grid(dirs, lat, sc)
    wts = torch.randn(U, D, 3, generator=g).to(device)
    (got * wts).sum().backward()
    p = attn_params(f.network)
    lat64 = lat.detach().cpu().double().requires_grad_(True)
    ref = O.reni_attention_decode(lat64[:, None].expand(U, D, L, 3).reshape(-1, L, 3), dirs.cpu().double()[None].expand(U, D, 3).reshape(-1, 3),
                                  sc.cpu().double()[:, None].expand(U, D).reshape(-1), p).reshape(U, D, 3)
    (ref * wts.cpu().double()).sum().backward()
    assert (got.detach().cpu().double() - ref.detach()).abs().max() < 2e-5 * ref.abs().max()
    assert (lat.grad.cpu().double() - lat64.grad).abs().max() < 1e-4 * lat64.grad.abs().max()
    # the per-pair entry point (background rays: one direction per latent) is the same function
    pp = f(dirs[None].expand(U, D, 3).reshape(-1, 3), lat.detach()[:, None].expand(U, D, L, 3).reshape(-1, L, 3), sc[:, None].expand(U, D).reshape(-1))
    assert (pp.reshape(U, D, 3) - got.detach()).abs().max() < 2e-5 * got.detach().abs().max()


@pytest.mark.gpu
def test_attention_decoder_matches_the_oracle_gpu():
    _check("cuda:0")          # a handful of directions per camera: the batched-product form


@pytest.mark.gpu
def test_attention_decoder_on_the_core_kernels():
    """D = 70 directions per camera: the attention core runs on csrc/attention.hip (partial workgroups, partial row blocks).  Values
    against the float64 oracle; latent gradients against the SAME decoder with the batched-product core on the same device (1e-5) --
    against float64 only to 5e-3: among the 4 x 70 x 256 x 6 ReLU units of the feed-forward blocks a pre-activation of ~1e-8 takes the
    other branch in float32 (measured: 7.4e-4 of the gradient maximum, identical for both cores and with every Linear / LayerNorm
    evaluated in float64)."""
    from neusky_amd import ops
    dev, L, U, D = "cuda:0", 12, 4, 70
    res = {}
    kernel_apply = ops.AttnCoreFn.apply
    try:
        for mode in ("kernel", "products"):
            if mode == "products":
                ops.AttnCoreFn.apply = staticmethod(lambda Q, dirs, Kt, Vt, scale, rd=None, rp=None, rs=None: _attn_core_reference(
                    Q.reshape(*dirs.shape[:2], -1), dirs, Kt, Vt, scale).reshape(Q.shape))
            f = _field(L, dev)
            g = torch.Generator().manual_seed(1)
            lat = (torch.randn(U, L, 3, generator=g) * 0.6).to(dev).requires_grad_(True)
            dirs = torch.randn(D, 3, generator=g)
            dirs = (dirs / dirs.norm(dim=-1, keepdim=True)).to(dev)
            sc = (torch.rand(U, generator=g) + 0.5).to(dev)
            got = f.forward_grid(dirs, lat, sc)
            wts = torch.randn(U, D, 3, generator=g).to(dev)
            (got * wts).sum().backward()
            res[mode] = (got.detach().cpu().double(), lat.grad.cpu().double())
    finally:
        ops.AttnCoreFn.apply = kernel_apply
    p = attn_params(f.network)
    lat64 = lat.detach().cpu().double().requires_grad_(True)
    ref = O.reni_attention_decode(lat64[:, None].expand(U, D, L, 3).reshape(-1, L, 3), dirs.cpu().double()[None].expand(U, D, 3).reshape(-1, 3),
                                  sc.cpu().double()[:, None].expand(U, D).reshape(-1), p).reshape(U, D, 3)
    (ref * wts.cpu().double()).sum().backward()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()  # noqa: E731
    assert rel(res["kernel"][0], ref.detach()) < 2e-5
    assert rel(res["kernel"][0], res["products"][0]) < 1e-5 and rel(res["kernel"][1], res["products"][1]) < 1e-5
    assert rel(res["kernel"][1], lat64.grad) < 5e-3


def _grid_and_rays(dev, D, R=37, one_camera=False):
    """forward_grid_and_rays (the rays as extra directions of their cameras: keys / values once per camera) against the two separate
    decodes it replaces -- forward_grid, and forward on the per-ray gathered latents -- values and latent / scale gradients"""
    L, U = 12, 5
    f = _field(L, dev)
    g = torch.Generator().manual_seed(11)
    lat0 = (torch.randn(U, L, 3, generator=g) * 0.6).to(dev)
    sc0 = (torch.rand(U, generator=g) + 0.5).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(D, 3, generator=g), dim=-1).to(dev)
    rdirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    cam = torch.randint(0, U, (R,), generator=g).to(dev)
    cam[:3] = 2  # several rays of one camera; camera 4 may have none
    if one_camera:  # a batch drawn from one image (the eval-latent fit): the camera's rays are walked by several workgroups
        cam[:] = 3
    wg, wr = torch.randn(U, D, 3, generator=g).to(dev), torch.randn(R, 3, generator=g).to(dev)
    res = []
    for mode in ("joint", "separate"):
        lat, sc = lat0.clone().requires_grad_(True), sc0.clone().requires_grad_(True)
        if mode == "joint":
            cols, bg = f.forward_grid_and_rays(dirs, lat, sc, rdirs, cam)
        else:
            cols, bg = f.forward_grid(dirs, lat, sc), f(rdirs, lat[cam], sc[cam])
        ((cols * wg).sum() + (bg * wr).sum()).backward()
        res.append((cols.detach(), bg.detach(), lat.grad.clone(), sc.grad.clone()))
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()  # noqa: E731
    for name, a, b in zip(("grid", "rays", "d latents", "d scale"), res[0], res[1]):
        assert rel(a, b) < 2e-5, (name, rel(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("D,R,one_camera", [(9, 37, False), (70, 37, False), (70, 150, True), (70, 1100, True)])
def test_rays_ride_with_their_cameras_gpu(D, R, one_camera):
    # D = 70: the grid rows on the matrix-core kernels, the rays' rows on the per-camera ray kernels (one camera: 32-ray slices, atomics)
    _grid_and_rays("cuda:0", D, R, one_camera)


@pytest.mark.gpu
@pytest.mark.parametrize("M,W,with_r", [(1001, 128, True), (1001, 128, False), (300, 64, True), (517, 256, True), (259, 512, True)])
def test_add_layer_norm_kernels_match_float64(M, W, with_r):
    """nsky_add_layer_norm_fwd / _bwd against torch in float64: sum, normalised rows, and the gradient of x and r with both outputs used"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(M + W)
    x, r = torch.randn(M, W, generator=g, dtype=torch.float64) * 2 + 0.3, torch.randn(M, W, generator=g, dtype=torch.float64)
    ln = torch.nn.LayerNorm(W).double()
    with torch.no_grad():
        ln.weight.add_(torch.randn(W, generator=g, dtype=torch.float64) * 0.2); ln.bias.add_(torch.randn(W, generator=g, dtype=torch.float64) * 0.2)
    ws, wy = torch.randn(M, W, generator=g, dtype=torch.float64), torch.randn(M, W, generator=g, dtype=torch.float64)
    x64, r64 = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    s64 = x64 + r64 if with_r else x64 * 1.0
    ((s64 * ws).sum() + (ln(s64) * wy).sum()).backward()
    dev = "cuda:0"
    lnf = torch.nn.LayerNorm(W).to(dev)
    lnf.load_state_dict({k: v.float() for k, v in ln.state_dict().items()})
    for p_ in lnf.parameters():
        p_.requires_grad_(False)
    xf, rf = x.float().to(dev).requires_grad_(True), r.float().to(dev).requires_grad_(True)
    s, y = ops.add_layer_norm(xf, rf if with_r else None, lnf)
    ((s * ws.float().to(dev)).sum() + (y * wy.float().to(dev)).sum()).backward()
    rel = lambda a, b: ((a.detach().cpu().double() - b).abs().max() / b.abs().max()).item()  # noqa: E731
    assert rel(s, s64.detach()) < 1e-6 and rel(y, ln(s64).detach()) < 2e-6
    assert rel(xf.grad, x64.grad) < 3e-6
    if with_r:
        assert rel(rf.grad, r64.grad) < 3e-6


@pytest.mark.gpu
def test_so2_equivariance_about_z():
    """rotating latents and directions together about z leaves the decoded radiance unchanged (RENI++'s defining property for
    equivariance="SO2", axis_of_invariance="z"); a rotation about another axis does not"""
    dev = "cuda:0"
    f = _field(10, dev)
    g = torch.Generator().manual_seed(3)
    lat = (torch.randn(3, 10, 3, generator=g) * 0.5).to(dev)
    dirs = torch.randn(17, 3, generator=g)
    dirs = (dirs / dirs.norm(dim=-1, keepdim=True)).to(dev)
    sc = torch.ones(3, device=dev)
    base = f.forward_grid(dirs, lat, sc)
    for a in (0.3, 2.1, -1.7):
        R = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]], device=dev)
        assert (f.forward_grid((dirs @ R.T).contiguous(), (lat @ R.T).contiguous(), sc) - base).abs().max() < 1e-5 * base.abs().max()
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, math.cos(0.5), -math.sin(0.5)], [0.0, math.sin(0.5), math.cos(0.5)]], device=dev)
    assert (f.forward_grid((dirs @ Rx.T).contiguous(), (lat @ Rx.T).contiguous(), sc) - base).abs().max() > 1e-3 * base.abs().max()
    # and the oracle has the same property
    lat, dirs = lat.cpu(), dirs.cpu()
    p = attn_params(f.network)
    B = 5
    l64, d64 = lat[0][None].expand(B, 10, 3).double(), dirs[:B].double()
    R = torch.tensor([[math.cos(0.9), -math.sin(0.9), 0.0], [math.sin(0.9), math.cos(0.9), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float64)
    a_, b_ = O.reni_attention_decode(l64, d64, torch.ones(B, dtype=torch.float64), p), O.reni_attention_decode(l64 @ R.T, d64 @ R.T, torch.ones(B, dtype=torch.float64), p)
    assert (a_ - b_).abs().max() < 1e-10 * a_.abs().max()


def test_reference_configuration_constructs():
    """neusky/configs/neusky_config.py:78-95 as written: Attention / VN / SO2 / z, latent 100, 8 heads x 6 layers, hidden 128, fixed decoder"""
    from neusky_amd.model_components.illumination import AttentionDecoder, RENIFieldConfig
    f = RENIFieldConfig(conditioning="Attention", invariant_function="VN", equivariance="SO2", axis_of_invariance="z", latent_dim=100,
                        hidden_features=128, num_attention_heads=8, num_attention_layers=6, fixed_decoder=True).setup()
    assert isinstance(f.network, AttentionDecoder) and len(f.network.layers) == 6 and not any(p.requires_grad for p in f.network.parameters())
    with pytest.raises(NotImplementedError):
        RENIFieldConfig(conditioning="Concat").setup()


@pytest.mark.gpu
def test_train_step_with_the_attention_decoder():
    """the reference's configured conditioning inside a full train iteration: finite losses, gradients reach the illumination latents,
    nothing reaches the frozen decoder"""
    from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
    from util_step import randomise, small_pipeline_config
    torch.manual_seed(0)
    cfg = small_pipeline_config(R=32, num_prop=(24, 12), S=8, D=24, images=4)
    cfg.model.illumination_field.conditioning = "Attention"
    pipe = cfg.setup(device="cuda:0")
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    lat0 = pipe.model.train_illumination_latents.detach().clone()
    for step in range(2):
        loss, ld, _ = train_iteration(pipe, opt, 10_000 + step)
        assert torch.isfinite(loss), ld
    assert not torch.equal(pipe.model.train_illumination_latents.detach(), lat0)
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in pipe.model.illumination_field.network.parameters())


def _attn_core_reference(Q, dirs, Kt, Vt, scale):
    """the batched-product form of the attention core (float64): q~ = scale [d_x q | d_y q | q], softmax(q~ K~^T) V~, combine"""
    U, D, H = Q.shape
    nh, L = Kt.shape[1], Kt.shape[2]
    dh = H // nh
    cx = torch.stack([dirs[..., 0], dirs[..., 1], torch.ones_like(dirs[..., 0])], -1).reshape(U, 1, D, 3, 1)
    Qh = (Q * scale).reshape(U, D, nh, 1, dh).transpose(1, 2)
    Qt = (Qh * cx).reshape(U, nh, D, 3 * dh)
    P = torch.softmax(Qt @ Kt.transpose(-1, -2), -1)
    O3 = (P @ Vt).reshape(U, nh, D, 3, dh)
    return (O3 * cx).sum(3).transpose(1, 2).reshape(U, D, H)


@pytest.mark.gpu
@pytest.mark.parametrize("U,D,L,nh", [(3, 512, 100, 8), (2, 77, 36, 8), (5, 33, 128, 2), (1, 1, 4, 1), (4, 64, 12, 8), (4, 70, 12, 8)])
def test_attention_core_kernels_match_float64(U, D, L, nh):
    """nsky_attn_core_fwd / _bwd (csrc/attention.hip) against the float64 batched-product form and its autograd: output, dQ and the
    per-camera dK~, dV~ (summed over the camera's rows); ragged D (partial workgroups, partial row blocks of the token kernel)"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(U * 1000 + D)
    H = 16 * nh
    Q = torch.randn(U, D, H, generator=g, dtype=torch.float64)
    dirs = torch.nn.functional.normalize(torch.randn(U, D, 3, generator=g, dtype=torch.float64), dim=-1)
    Kt = torch.randn(U, nh, L, 48, generator=g, dtype=torch.float64) * 0.7
    Vt = torch.randn(U, nh, L, 48, generator=g, dtype=torch.float64)
    dO = torch.randn(U, D, H, generator=g, dtype=torch.float64)
    ref_in = [t.clone().requires_grad_(True) for t in (Q, Kt, Vt)]
    ref = _attn_core_reference(ref_in[0], dirs, ref_in[1], ref_in[2], 0.25)
    ref.backward(dO)
    dev = "cuda:0"
    got_in = [t.float().to(dev).requires_grad_(True) for t in (Q, Kt, Vt)]
    got = ops.AttnCoreFn.apply(got_in[0], dirs.float().to(dev), got_in[1], got_in[2], 0.25)
    got.backward(dO.float().to(dev))
    rel = lambda a, b: (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)  # noqa: E731
    assert rel(got, ref.detach()) < 5e-6
    for name, a, b in zip(("dQ", "dK~", "dV~"), got_in, ref_in):
        assert rel(a.grad, b.grad) < 2e-5, (name, rel(a.grad, b.grad))


@pytest.mark.gpu
def test_frozen_feed_forward_node_matches_two_dense_layers():
    """ops.FrozenFeedForwardFn (ReLU mask in the second layer's input-gradient epilogue) against the two-layer composition it replaces"""
    from neusky_amd import ops
    dev, M = "cuda:0", 5000
    g = torch.Generator().manual_seed(2)
    ff1, ff2 = torch.nn.Linear(128, 256).to(dev), torch.nn.Linear(256, 128).to(dev)
    for p_ in list(ff1.parameters()) + list(ff2.parameters()):
        p_.requires_grad_(False)
    x0 = torch.randn(M, 128, generator=g).to(dev)
    w = torch.randn(M, 128, generator=g).to(dev)
    outs = []
    for mode in ("node", "layers"):
        x = x0.clone().requires_grad_(True)
        if mode == "node":
            y = ops.FrozenFeedForwardFn.apply(x, ff1.weight, ff1.bias, ff2.weight, ff2.bias)
        else:
            y = ops.DenseFn.apply(ops.DenseFn.apply(x, ff1.weight, ff1.bias, 256, "relu", True), ff2.weight, ff2.bias, 128, "none", True)
        (y * w).sum().backward()
        outs.append((y.detach(), x.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and (outs[0][1] - outs[1][1]).abs().max().item() <= 1e-6 * outs[1][1].abs().max().item()
    ref = torch.nn.functional.linear(torch.relu(torch.nn.functional.linear(x0.double(), ff1.weight.double(), ff1.bias.double())), ff2.weight.double(), ff2.bias.double())
    assert ((outs[0][0].double() - ref).abs().max() / ref.abs().max()).item() < 2e-6


def _one_camera(dev, B):
    f = _field(12, dev)
    g = torch.Generator().manual_seed(5)
    lat = (torch.randn(12, 3, generator=g) * 0.6).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(dev)
    sc = torch.tensor(1.3, device=dev)
    c, s = math.cos(0.7), math.sin(0.7)
    rot = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], device=dev)
    for r in (None, rot):
        a = f.forward_camera(dirs, lat, sc, r)
        b = f(dirs, lat[None].expand(B, -1, -1), sc.expand(B), r)
        assert ((a - b).abs().max() / b.abs().max()).item() < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("D", [17, 300])  # short direction lists: the vector-unit kernels; the render chunk's form: the matrix-core kernels
def test_one_camera_decode_equals_the_per_direction_decode_gpu(D):
    _one_camera("cuda:0", D)


@pytest.mark.gpu
def test_attention_decoder_at_the_bench_size_against_the_oracle():
    """The decode of the bench key `attention_decoder` at its own size -- 300 cameras x 512 directions + 1024 ray rows, 100 tokens, the
    matrix-core attention kernels with their real grids, the per-camera ray kernels on a real ray-to-camera distribution, fused
    add + layer norm, the feed-forward node -- against the float64 oracle on 40 (camera, direction) pairs and 24 rays spread over
    the batch, and the latent gradient of a loss on exactly those rows."""
    dev, U, D, L, R = "cuda:0", 300, 512, 100, 1024
    f = _field(L, dev)
    g = torch.Generator().manual_seed(9)
    lat = (torch.randn(U, L, 3, generator=g) * 0.6).to(dev).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(D, 3, generator=g), dim=-1).to(dev)
    sc = (torch.rand(U, generator=g) + 0.5).to(dev)
    rdirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    cam = torch.randint(0, U, (R,), generator=g).to(dev)
    cols, bg = f.forward_grid_and_rays(dirs, lat, sc, rdirs, cam)
    assert cols.shape == (U, D, 3) and bg.shape == (R, 3) and torch.isfinite(cols).all() and torch.isfinite(bg).all()
    pu = torch.cat([torch.tensor([0, 0, 299, 299, 150]), torch.randint(0, U, (35,), generator=g)])
    pd = torch.cat([torch.tensor([0, 511, 0, 511, 255]), torch.randint(0, D, (35,), generator=g)])
    pr = torch.cat([torch.tensor([0, R - 1]), torch.randint(0, R, (22,), generator=g)])
    wc, wb = torch.randn(40, 3, generator=g), torch.randn(24, 3, generator=g)
    ((cols[pu.to(dev), pd.to(dev)] * wc.to(dev)).sum() + (bg[pr.to(dev)] * wb.to(dev)).sum()).backward()
    p = attn_params(f.network)
    lat64 = lat.detach().cpu().double().requires_grad_(True)
    camc = cam.cpu()
    ref_c = O.reni_attention_decode(lat64[pu], dirs.cpu().double()[pd], sc.cpu().double()[pu], p)
    ref_b = O.reni_attention_decode(lat64[camc[pr]], rdirs.cpu().double()[pr], sc.cpu().double()[camc[pr]], p)
    ((ref_c * wc.double()).sum() + (ref_b * wb.double()).sum()).backward()
    rel = lambda a, b: ((a.detach().cpu().double() - b.detach()).abs().max() / b.detach().abs().max()).item()  # noqa: E731
    assert rel(cols[pu.to(dev), pd.to(dev)], ref_c) < 2e-5 and rel(bg[pr.to(dev)], ref_b) < 2e-5
    assert rel(lat.grad, lat64.grad) < 2e-3  # (ReLU units of the feed-forward blocks near zero: see test_attention_decoder_on_the_core_kernels)
