"""ops.RayReduceFn / ops.NormalizeFn against the torch expressions they replace (neusky_model.py:591-595, :812-813, :1342-1357;
sdf_albedo_field.py:256), values and gradients in float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(w, st, en, nr, al, max_clamp):
    steps = (st + en) / 2
    depth = torch.sum(w * steps, dim=-2) / (torch.sum(w, -2) + 1e-10)
    depth = torch.clip(depth, steps.min(), steps.max())
    if max_clamp > 0:
        depth = torch.clamp(depth, max=max_clamp)
    acc = w.sum(dim=-2)
    normal = torch.sum(w * nr, dim=-2) if nr is not None else None
    alb = torch.sum(w * al, dim=-2) + (1.0 - w.sum(dim=-2)) if al is not None else None
    return depth, acc, normal, alb


@pytest.mark.parametrize("R,S,with_n,with_a,max_clamp", [(37, 96, True, True, 0.0), (1024, 96, True, False, 3.0), (5, 130, False, False, 0.0)])
def test_ray_reduce_matches_torch(R, S, with_n, with_a, max_clamp):
    from neusky_amd import ops
    g = torch.Generator().manual_seed(R)
    w = torch.rand(R, S, 1, generator=g) * 0.05
    w[0] = 0.0  # an empty ray: its depth is clipped up to the smallest mid point (no gradient to the weights through the depth)
    w[1] *= 60.0 / S  # a ray whose expected depth exceeds max_clamp
    st = torch.cumsum(torch.rand(R, S, 1, generator=g) * 0.1, dim=1) + 0.05
    en = st + 0.05
    nr = torch.nn.functional.normalize(torch.randn(R, S, 3, generator=g), dim=-1) if with_n else None
    al = torch.rand(R, S, 3, generator=g) if with_a else None
    leaf = lambda t: None if t is None else t.to(DEV).requires_grad_(True)  # noqa: E731
    wd, nd, ad = leaf(w), leaf(nr), leaf(al)
    p2p, acc, normal, alb = ops.RayReduceFn.apply(wd, st.to(DEV), en.to(DEV), nd, ad, max_clamp)
    w64 = w.double().requires_grad_(True)
    n64 = None if nr is None else nr.double().requires_grad_(True)
    a64 = None if al is None else al.double().requires_grad_(True)
    rp, ra, rn, rb = _ref(w64, st.double(), en.double(), n64, a64, max_clamp)
    assert torch.allclose(p2p.cpu().double(), rp, atol=2e-6, rtol=1e-5) and torch.allclose(acc.cpu().double(), ra, atol=2e-6, rtol=1e-5)
    gp, ga = torch.randn(R, 1, generator=g), torch.randn(R, 1, generator=g)
    loss = (p2p * gp.to(DEV)).sum() + (acc * ga.to(DEV)).sum()
    ref = (rp * gp.double()).sum() + (ra * ga.double()).sum()
    if with_n:
        gn = torch.randn(R, 3, generator=g)
        assert torch.allclose(normal.cpu().double(), rn, atol=2e-6, rtol=1e-5)
        loss = loss + (normal * gn.to(DEV)).sum(); ref = ref + (rn * gn.double()).sum()
    if with_a:
        gb = torch.randn(R, 3, generator=g)
        assert torch.allclose(alb.cpu().double(), rb, atol=2e-6, rtol=1e-5)
        loss = loss + (alb * gb.to(DEV)).sum(); ref = ref + (rb * gb.double()).sum()
    loss.backward(); ref.backward()
    scale = float(w64.grad.abs().max())
    assert float((wd.grad.cpu().double() - w64.grad).abs().max()) < 2e-5 * scale
    if with_n:
        assert torch.allclose(nd.grad.cpu().double(), n64.grad, atol=1e-6, rtol=1e-5)
    if with_a:
        assert torch.allclose(ad.grad.cpu().double(), a64.grad, atol=1e-6, rtol=1e-5)


def test_normalize_matches_torch():
    from neusky_amd import ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn(5000, 3, generator=g) * torch.exp(torch.randn(5000, 1, generator=g) * 3)
    x[7] = 0.0
    xd = x.to(DEV).requires_grad_(True)
    n = ops.NormalizeFn.apply(xd)
    x64 = x.double().requires_grad_(True)
    r = torch.nn.functional.normalize(x64, p=2, dim=-1)
    assert torch.allclose(n.cpu().double(), r, atol=1e-6)
    gn = torch.randn(5000, 3, generator=g)
    gn[7] = 0.0  # (the clamped branch's gradient is g / eps: kept finite here)
    (n * gn.to(DEV)).sum().backward(); (r * gn.double()).sum().backward()
    scale = x64.grad.abs().max(dim=-1, keepdim=True).values + 1e-30
    assert float(((xd.grad.cpu().double() - x64.grad).abs() / scale).max()) < 1e-4


def test_term_points_match_torch_autograd():
    """ops.TermPointsFn (ddf_model.py:243 / neusky_model.py:1716-1724) against the torch expression it replaces, values and both gradients"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(8)
    R, Dv, N = 37, 19, 55
    M = R * Dv
    sp = torch.randn(M, 3, generator=g)
    sel = torch.nn.functional.normalize(torch.randn(Dv, 3, generator=g), dim=-1)
    pos, dirs = torch.randn(N, 3, generator=g), torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
    t_hat, t_main = torch.rand(M, generator=g), torch.rand(N, generator=g)
    th, tm = t_hat.to(DEV).requires_grad_(True), t_main.to(DEV).requires_grad_(True)
    out = ops.TermPointsFn.apply(sp.to(DEV), sel.to(DEV), th, pos.to(DEV), dirs.to(DEV), tm)
    th64, tm64 = t_hat.double().requires_grad_(True), t_main.double().requires_grad_(True)
    wd = (-sel.double())[None].expand(R, Dv, 3).reshape(-1, 3)
    ref = torch.cat([sp.double() + wd * th64[:, None], pos.double() + dirs.double() * tm64[:, None]], 0)
    assert torch.allclose(out.cpu().double(), ref, atol=1e-6)
    probe = torch.randn(M + N, 3, generator=g)
    (out * probe.to(DEV)).sum().backward(); (ref * probe.double()).sum().backward()
    assert torch.allclose(th.grad.cpu().double(), th64.grad, atol=1e-5) and torch.allclose(tm.grad.cpu().double(), tm64.grad, atol=1e-5)
    # visibility rows only
    out2 = ops.TermPointsFn.apply(sp.to(DEV), sel.to(DEV), th.detach(), None, None, None)
    assert torch.equal(out2, out[:M].detach())


def test_sigmoid_column_matches_torch():
    """ops.SigmoidColumnFn (directional_distance_field.py:297-299) against torch.sigmoid(raw[:, 0]) * scale, values and gradient"""
    from neusky_amd import ops
    g = torch.Generator().manual_seed(31)
    raw = torch.randn(5001, 4, generator=g) * 3
    probe = torch.randn(5001, generator=g)
    a = raw.to(DEV).requires_grad_(True)
    t = ops.SigmoidColumnFn.apply(a, 2.5)
    (t * probe.to(DEV)).sum().backward()
    b = raw.double().requires_grad_(True)
    ref = torch.sigmoid(b[:, 0]) * 2.5
    (ref * probe.double()).sum().backward()
    assert torch.allclose(t.detach().cpu().double(), ref.detach(), atol=1e-6)
    assert torch.allclose(a.grad.cpu().double(), b.grad, atol=1e-6) and bool((a.grad[:, 1:] == 0).all())
