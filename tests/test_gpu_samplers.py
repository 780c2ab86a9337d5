"""The fused sample generators (csrc/samplers.hip) against the distributions of the reference's host samplers (their RNG
streams are not portable; the oracle's restatement of the same algorithm gives the numbers to compare with)."""
import math

import pytest
import torch

from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sampler(n_pos, n_dir, kappa=20.0):
    from neusky_amd.model_components.ddf_sampler import VMFDDFSampler, VMFDDFSamplerConfig
    return VMFDDFSampler(VMFDDFSamplerConfig(num_samples_on_sphere=n_pos, num_rays_per_sample=n_dir, concentration=kappa), device=DEV)


def test_vmf_ddf_samples_distribution():
    """ddf_sampler.py:205-286: unit positions on the upper hemisphere, unit directions into the inward half space, and the lobe's
    concentration = that of the reference algorithm (the oracle's restatement, G10)"""
    s = _sampler(64, 2000)
    rb = s()
    o, d = rb.origins.view(64, 2000, 3).cpu(), rb.directions.view(64, 2000, 3).cpu()
    assert torch.allclose(o.norm(dim=-1), torch.ones(64, 2000), atol=1e-5) and bool((o[..., 2] >= 0).all())
    assert bool((o[:, :1] == o).all()), "a position's directions share its origin"
    assert torch.allclose(d.norm(dim=-1), torch.ones(64, 2000), atol=1e-5)
    cos = (d * -o).sum(-1)
    assert bool((cos >= 0).all())
    P, D = O.vmf_ddf_rays(16, 4000, 20.0, 1.0, torch.Generator().manual_seed(1))
    ref = (D.view(16, 4000, 3) * -P.view(16, 4000, 3)).sum(-1)
    assert abs(float(cos.mean()) - float(ref.mean())) < 0.003, (float(cos.mean()), float(ref.mean()))
    assert abs(float(cos.std()) - float(ref.std())) < 0.004
    # positions: uniform on the (folded) sphere -> z uniform on [0, 1], azimuth uniform
    z = o[:, 0, 2]
    big = _sampler(20000, 1)().origins.cpu()
    assert abs(float(big[:, 2].mean()) - 0.5) < 0.01 and abs(float(big[:, 2].var()) - 1 / 12) < 0.005
    az = torch.atan2(big[:, 1], big[:, 0])
    assert abs(float(az.mean())) < 0.05 and abs(float(az.var()) - math.pi**2 / 3) < 0.1
    # the tangential part of a direction is uniform around the normal: no preferred azimuth in a fixed world frame
    t = d - cos[..., None] * -o
    assert float(t.mean(dim=(0, 1)).abs().max()) < 0.01
    assert z.numel() == 64


def test_vmf_ddf_samples_advance_and_reproduce():
    s = _sampler(8, 16)
    from neusky_amd.utils.utils import device_rng
    a = s()
    b = s()
    assert not torch.equal(a.directions, b.directions) and not torch.equal(a.origins, b.origins)
    _, counter = device_rng(s, "ddf_vmf_samples", 0, DEV)  # the generator's call counter lives on the device (checkpointed, graph-safe)
    assert int(counter) == 2
    counter.zero_()
    c = s()
    assert torch.equal(a.directions, c.directions) and torch.equal(a.origins, c.origins)


def test_vmf_ddf_samples_replay_in_a_hip_graph_draw_fresh_numbers():
    s = _sampler(8, 16)
    s()  # state tensors exist before capture
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g):
            rb = s()
        g.replay(); first = rb.directions.clone()
        g.replay(); second = rb.directions.clone()
    torch.cuda.synchronize()
    assert not torch.equal(first, second)


def test_reni_grid_inputs_match_the_torch_construction():
    """ops.RENIGridInputsFn against RENIField.forward_grid's torch construction of the same two matrices (values and d latents)"""
    from neusky_amd import ops
    from neusky_amd.fields.directional_distance_field import nerf_encoding
    g = torch.Generator().manual_seed(2)
    U, L, D = 7, 100, 512
    Z = torch.randn(U, L, 3, generator=g)
    dirs = torch.nn.functional.normalize(torch.randn(D, 3, generator=g), dim=-1)
    Zd = Z.to(DEV).requires_grad_(True)
    cond, x = ops.RENIGridInputsFn.apply(Zd, dirs.to(DEV))
    # ... and with ray rows behind the grid rows: the same rows as a one-direction-per-latent gather
    R = 333
    rd = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    rl = torch.randint(0, U, (R,), generator=g)
    Zr = Z.to(DEV).requires_grad_(True)
    cond2, x2 = ops.RENIGridInputsFn.apply(Zr, dirs.to(DEV), rd.to(DEV), rl.to(DEV))
    assert torch.equal(cond2[:U * D], cond) and torch.equal(x2[:U * D], x)
    from neusky_amd.model_components.illumination import RENIField
    Zq = Z.double().requires_grad_(True)
    rc2, rx2 = RENIField.invariant_inputs(Zq[rl], rd.double())
    assert torch.allclose(cond2[U * D:].cpu().double(), rc2, atol=2e-6, rtol=1e-6) and torch.allclose(x2[U * D:, :2].cpu().double(), rx2, atol=1e-6)
    gr = torch.randn(R, 300, generator=g)
    (cond2[U * D:] * gr.to(DEV)).sum().backward(); (rc2 * gr.double()).sum().backward()
    assert torch.allclose(Zr.grad.cpu().double(), Zq.grad, atol=1e-4, rtol=1e-5)
    Z64 = Z.double().requires_grad_(True)
    d64 = dirs.double()
    zxy, zz = Z64[..., :2], Z64[..., 2]
    dxy, dz = d64[:, :2], d64[:, 2]
    dot = torch.einsum("uln,dn->udl", zxy, dxy)
    rc = torch.stack([zxy.norm(dim=-1)[:, None, :].expand(U, D, L), zz[:, None, :].expand(U, D, L), dot], -1).reshape(U * D, 3 * L)
    rx = torch.stack([dxy.norm(dim=-1), dz], -1)[None].expand(U, D, 2).reshape(U * D, 2)
    rx = torch.cat([rx, nerf_encoding(rx, 2, 2.0)], -1)
    assert cond.shape == (U * D, 300) and x.shape == (U * D, 12)
    assert torch.allclose(cond.cpu().double(), rc, atol=2e-6, rtol=1e-6), float((cond.cpu().double() - rc).abs().max())
    assert torch.allclose(x.cpu().double()[:, :10], rx, atol=1e-5), float((x.cpu().double()[:, :10] - rx).abs().max())
    assert bool((x[:, 10:] == 0).all())
    gc = torch.randn(U * D, 300, generator=g)
    (cond * gc.to(DEV)).sum().backward()
    (rc * gc.double()).sum().backward()
    assert torch.allclose(Zd.grad.cpu().double(), Z64.grad, atol=1e-4, rtol=1e-5)


def _illumination_sampler(num_directions=512, icosphere_order=None):
    from neusky_amd.model_components.illumination import IcosahedronSampler, IcosahedronSamplerConfig
    kw = dict(num_directions=num_directions) if icosphere_order is None else dict(icosphere_order=icosphere_order)
    return IcosahedronSampler(IcosahedronSamplerConfig(apply_random_rotation=True, **kw))


def test_illumination_directions_kernel_with_a_given_rotation_matches_host_path():
    """illumination_samplers.py:75-110 + neusky_model.py:1650-1657: rotated set and its z > 0 subset (ascending indices)"""
    from neusky_amd.model_components.illumination import random_rotation
    s = _illumination_sampler(512)
    for seed in range(4):
        R = random_rotation(torch.Generator().manual_seed(seed)).float()
        dirs, sel = s.on_device(DEV, rotation=R)
        want = s.directions.double() @ R.double().T
        assert torch.allclose(dirs.cpu().double(), want, atol=2e-7)
        want_sel = torch.nonzero(want[:, 2] > 0)[:, 0]
        assert sel.dtype == torch.int32 and torch.equal(sel.cpu().long(), want_sel)
        assert torch.allclose(s.last_rotation.cpu(), R)


def test_illumination_directions_kernel_draws_uniform_rotations():
    torch.manual_seed(11)
    s = _illumination_sampler(512)
    base = s.directions.double()
    rots = []
    for _ in range(200):
        dirs, sel = s.on_device(DEV)
        R = s.last_rotation.cpu().double()
        rots.append(R)
        assert torch.allclose(R @ R.T, torch.eye(3, dtype=torch.float64), atol=1e-5) and abs(float(torch.det(R)) - 1.0) < 1e-5
        assert torch.allclose(dirs.cpu().double(), base @ R.T, atol=2e-6)
        z = dirs[:, 2].cpu()
        assert sel.numel() == 256 and bool((z[sel.cpu().long()] > 0).all()) and int((z > 0).sum()) == 256
        assert bool((sel[1:] > sel[:-1]).all())
    R = torch.stack(rots)
    assert (R[1:] - R[:-1]).abs().amax(dim=(1, 2)).min() > 1e-3, "successive calls draw different rotations"
    # Haar measure: every matrix entry has mean 0 and variance 1/3; the image of e_z is uniform on the sphere
    assert R.mean(0).abs().max() < 0.15 and (R.var(0) - 1.0 / 3.0).abs().max() < 0.1
    # same seed, same call number -> same rotation (counter-based generator)
    torch.manual_seed(11)
    s2 = _illumination_sampler(512)
    s2.on_device(DEV)
    assert torch.equal(s2.last_rotation.cpu().double(), rots[0])


def test_illumination_directions_kernel_icosphere_set():
    """icosphere order 2 (162 vertices, centrally symmetric): not a multiple of the wave size"""
    s = _illumination_sampler(icosphere_order=2)
    D = s.directions.shape[0]
    assert D % 2 == 0
    dirs, sel = s.on_device(DEV)
    z = dirs[:, 2].cpu()
    assert sel.numel() == D // 2 and torch.equal(sel.cpu().long(), torch.nonzero(z > 0)[:, 0])


def test_reni_output_matches_exp_times_scale():
    """ops.RENIOutputFn against neusky_model.py:488-549's exp + per-image scale (float64 autograd): values, d raw, d scale"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(4)
    U, D, R = 7, 100, 333  # 700 grid rows: a wave holds rows of two images
    raw = torch.randn(U * D + R, 4, generator=g)
    scale = torch.rand(U, generator=g) + 0.5
    rl = torch.randint(0, U, (R,), generator=g)
    rawd, sd = raw.to(DEV).requires_grad_(True), scale.to(DEV).requires_grad_(True)
    grid, rays = ops.RENIOutputFn.apply(rawd, sd, rl.to(DEV), U, D)
    r64, s64 = raw.double().requires_grad_(True), scale.double().requires_grad_(True)
    e = torch.exp(r64[:, :3])
    want_grid, want_rays = e[:U * D].reshape(U, D, 3) * s64[:, None, None], e[U * D:] * s64[rl][:, None]
    assert torch.allclose(grid.cpu().double(), want_grid, rtol=2e-6) and torch.allclose(rays.cpu().double(), want_rays, rtol=2e-6)
    gg, gr = torch.randn(U, D, 3, generator=g), torch.randn(R, 3, generator=g)
    ((grid * gg.to(DEV)).sum() + (rays * gr.to(DEV)).sum()).backward()
    ((want_grid * gg.double()).sum() + (want_rays * gr.double()).sum()).backward()
    assert torch.allclose(rawd.grad.cpu().double()[:, :3], r64.grad[:, :3], rtol=1e-5, atol=1e-6) and bool((rawd.grad[:, 3] == 0).all())
    assert torch.allclose(sd.grad.cpu().double(), s64.grad, rtol=1e-4, atol=1e-4)
    # only the rays' output used (the grid's gradient is absent, not zero-filled)
    rawd.grad = None; sd.grad = None
    grid, rays = ops.RENIOutputFn.apply(rawd, sd, rl.to(DEV), U, D)
    (rays * gr.to(DEV)).sum().backward()
    want = torch.autograd.grad(((torch.exp(r64[U * D:, :3]) * s64[rl][:, None]) * gr.double()).sum(), [r64, s64])
    assert torch.allclose(rawd.grad.cpu().double()[:, :3], want[0][:, :3], rtol=1e-5, atol=1e-6)
    assert torch.allclose(sd.grad.cpu().double(), want[1], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("n0,nb", [(256, 97), (96, 65), (100, 33), (7, 5)])
def test_pdf_sample_indices_bit_exact_against_float32_torch(n0, nb):
    """nsky_pdf_sample (nerfstudio PDFSampler via ProposalNetworkSampler, neusky_model.py:561): the searchsorted indices equal
    those of the float32 torch formulation (oracle.pdf_sample_bins run in float32: CPU cumsum order) bit for bit, the new bins to rounding"""
    from neusky_amd import hip
    g = torch.Generator().manual_seed(n0 * 1000 + nb)
    R = 300
    weights = torch.rand(R, n0, generator=g) ** 3
    weights[::7] *= 1e-4  # nearly empty rays
    bins = torch.sort(torch.rand(R, n0 + 1, generator=g), -1).values
    jitter = torch.rand(R, 1, generator=g)
    want_bins, want_inds = O.pdf_sample_bins(bins, weights, nb - 1, jitter)
    u = torch.linspace(0.0, 1.0 - (1.0 / nb), steps=nb)
    got_bins, got_inds = hip.pdf_sample(weights.to(DEV), bins.to(DEV), u.to(DEV), jitter.reshape(-1).to(DEV), nb, want_inds=True)
    assert torch.equal(got_inds.cpu().long(), want_inds)
    assert torch.allclose(got_bins.cpu(), want_bins, atol=1e-6)


def test_grid_probe_points_distribution():
    """neusky_model.py:704-724: jitter uniform inside each lattice cell, directions uniform on the sphere, fresh draws per call"""
    from neusky_amd import hip
    res = 10
    lin = torch.linspace(-1.0, 1.0, res)
    lattice = torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3).to(DEV)
    gap = [0.2, 0.2, 0.25]
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    pos, dirs = torch.empty_like(lattice), torch.empty_like(lattice)
    draws = []
    for _ in range(40):
        hip.grid_probe_points(lattice, gap, 1234, counter, pos, dirs)
        draws.append(((pos - lattice).cpu(), dirs.cpu().clone()))
    assert int(counter) == 40
    off = torch.stack([d[0] for d in draws])  # [40, 1000, 3]
    g = torch.tensor(gap)
    assert bool((off.abs() <= g / 2 + 1e-6).all())
    u = off / g + 0.5  # ~ U(0, 1)
    assert (u.mean((0, 1)) - 0.5).abs().max() < 0.01 and (u.var((0, 1)) - 1.0 / 12.0).abs().max() < 0.005
    d = torch.stack([d[1] for d in draws])
    assert torch.allclose(d.norm(dim=-1), torch.ones(40, 1000), atol=1e-5)
    assert d.mean((0, 1)).abs().max() < 0.02 and (d.var((0, 1)) - 1.0 / 3.0).abs().max() < 0.01
    assert (draws[0][0] - draws[1][0]).abs().max() > 1e-3
    # same seed and call number -> the same draw
    c2 = torch.zeros(1, dtype=torch.int64, device=DEV)
    p2, d2 = torch.empty_like(lattice), torch.empty_like(lattice)
    hip.grid_probe_points(lattice, gap, 1234, c2, p2, d2)
    assert torch.equal((p2 - lattice).cpu(), draws[0][0]) and torch.equal(d2.cpu(), draws[0][1])


@pytest.mark.parametrize("P,in_dim,ldf", [(5000, 10, 12), (262144, 10, 12), (777, 12, 12), (1500, 7, 8)])
def test_proposal_mlp_matches_torch(P, in_dim, ldf):
    """ops.ProposalMLPFn (nerfstudio HashMLPDensityField's Linear + ReLU -> Linear head) against torch in float64: values and every gradient"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(P + in_dim)
    feat = torch.zeros(P, ldf)
    feat[:, :in_dim] = torch.randn(P, in_dim, generator=g)
    lin0, lin1 = torch.nn.Linear(in_dim, 16), torch.nn.Linear(16, 1)
    with torch.no_grad():
        lin0.bias.uniform_(-0.5, 0.5)
    probe = torch.randn(P, 1, generator=g)
    fd = feat.to(DEV).requires_grad_(True)
    ps = [p.detach().clone().to(DEV).requires_grad_(True) for p in (lin0.weight, lin0.bias, lin1.weight, lin1.bias)]
    raw = ops.ProposalMLPFn.apply(fd, *ps)
    assert raw.shape == (P, 1)
    (raw * probe.to(DEV)).sum().backward()
    f64 = feat.double().requires_grad_(True)
    q = [p.detach().double().requires_grad_(True) for p in (lin0.weight, lin0.bias, lin1.weight, lin1.bias)]
    ref = torch.relu(f64[:, :in_dim] @ q[0].T + q[1]) @ q[2].T + q[3]
    (ref * probe.double()).sum().backward()
    assert torch.allclose(raw.detach().cpu().double(), ref.detach(), atol=2e-5)
    assert torch.allclose(fd.grad.cpu().double()[:, :in_dim], f64.grad[:, :in_dim], atol=1e-5) and bool((fd.grad[:, in_dim:] == 0).all())
    for a, b, n in zip(ps, q, ("dW0", "db0", "dW1", "db1")):
        tol = (2e-5 + 1e-9 * P) * max(1.0, float(b.grad.abs().max()))  # (fp32 sums of P terms, order not fixed)
        assert torch.allclose(a.grad.cpu().double(), b.grad, atol=tol), (n, float((a.grad.cpu().double() - b.grad).abs().max()))


def test_fit_rows_are_finite_on_the_pole_and_drawn_points_stay_off_it():
    """get_localised_transforms (ddf_model.py:158-181) divides 0 by 0 for a position on the z axis.  A multi-view point handed in exactly
    on the pole gives finite encoded rows (any unit vector across the axis completes the frame), and the kernel's own draws never land
    there: cos(phi) = 2 u - 1 with u strictly inside (0, 1) in float (round 4: u = 1.0 came out once in 2^24 draws -- a NaN in a
    3000-step training run)."""
    from neusky_amd import hip
    N = 4096
    g = torch.Generator().manual_seed(0)
    pos = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1).to(DEV)
    term = (torch.rand(N, generator=g) * 0.5 + 0.2).to(DEV).reshape(N, 1).contiguous()
    mv = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    mv[:7] = torch.tensor([0.0, 0.0, 1.0])
    mv[7:9] = torch.tensor([0.0, 0.0, -1.0])
    mv = mv.to(DEV).contiguous()
    q_pos, xrow = torch.full((2 * N, 3), float("nan"), device=DEV), torch.full((2 * N, 16), float("nan"), device=DEV)
    mv_out = torch.empty(N, 3, device=DEV)
    hip.ddf_fit_rows_fwd(pos, dirs, term.reshape(N), mv, 0, None, None, None, 1.0, True, 3.0, False, q_pos, xrow, mv_out, None, None)
    torch.cuda.synchronize()
    assert torch.isfinite(q_pos).all() and torch.isfinite(xrow[:, :15]).all()
    d_loc = xrow[N:N + 9, :3]
    assert ((d_loc.norm(dim=-1) - 1).abs() < 1e-5).all()  # a rotation of a unit direction
    # drawn points: |z| < 1 strictly, over 2^22 draws
    counter = torch.zeros(1, dtype=torch.int64, device=DEV)
    worst = 0.0
    for _ in range(1024):
        hip.ddf_fit_rows_fwd(pos, dirs, term.reshape(N), None, 1234, counter, None, None, 1.0, True, 3.0, False, q_pos, xrow, mv_out, None, None)
        worst = max(worst, float(mv_out[:, 2].abs().max()))
        assert torch.isfinite(xrow[:, :15]).all()
    assert worst < 1.0, worst
