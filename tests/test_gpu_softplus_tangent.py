"""nsky_softplus_tangent_bwd: the reverse-over-forward step of a Softplus layer carrying three input tangents (the double backward the
reference gets from torch.autograd.grad(create_graph=True), sdf_albedo_field.py:235-238), float4 and scalar forms, with the
weighted column sum of the tangents (the sdf row's gradient) from the same pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(da, s, ta, g, beta):
    """dz = da s + sum_k g_k ta_k beta (1 - s), du_k = g_k s"""
    k1 = beta * (1.0 - s)
    dz = (0 if da is None else da * s) + (g * ta * k1[None]).sum(0)
    return dz, g * s[None]


@pytest.mark.parametrize("N,C,ld", [(1000, 256, 256), (777, 64, 72), (513, 30, 32)])
@pytest.mark.parametrize("outer", [True, False])
def test_softplus_tangent_bwd_matches_float64(N, C, ld, outer):
    from neusky_amd import hip
    gen = torch.Generator().manual_seed(N + C)
    beta = 100.0
    s = torch.rand(N, ld, generator=gen)
    ta = torch.randn(3, N, ld, generator=gen)
    da = torch.randn(N, ld, generator=gen)
    ggrad = torch.randn(N, 3, generator=gen)
    wvec = torch.randn(C, generator=gen)
    dta = torch.randn(3, N, ld, generator=gen)
    d = lambda t: t.to(DEV)  # noqa: E731
    dz = torch.full((N, ld), float("nan"), device=DEV)
    du = torch.full((3, N, ld), float("nan"), device=DEV)
    wsum0 = torch.randn(C, generator=gen)
    wsum = d(wsum0.clone())
    if outer:
        hip.softplus_tangent_bwd(d(da), d(s), d(ta), None, d(ggrad), d(wvec), beta, N, C, dz, du, wsum=wsum)
        g = ggrad.double().t()[:, :, None] * wvec.double()[None, None, :]
    else:
        hip.softplus_tangent_bwd(None, d(s), d(ta), d(dta), None, None, beta, N, C, dz, du)
        g = dta.double()[:, :, :C]
    torch.cuda.synchronize()
    want_dz, want_du = _reference(da.double()[:, :C] if outer else None, s.double()[:, :C], ta.double()[:, :, :C], g, beta)
    assert torch.allclose(dz[:, :C].cpu().double(), want_dz, rtol=1e-5, atol=1e-4)
    assert torch.allclose(du[:, :, :C].cpu().double(), want_du, rtol=1e-5, atol=1e-5)
    if ld > C:
        assert bool(torch.isnan(dz[:, C:]).all()), "pad columns are not written"
    if outer:
        want = wsum0.double() + (ggrad.double().t()[:, :, None] * ta.double()[:, :, :C]).sum((0, 1))
        assert torch.allclose(wsum.cpu().double(), want, rtol=1e-4, atol=1e-3)
