"""The fused sdf value chain (nsky_sdf_pack + nsky_sdf_chain_fwd / _bwd: SDFAlbedoField.get_sdf_at_pos,
neusky/fields/sdf_albedo_field.py:169-174) against a float64 torch restatement of the same three layers and against the
per-layer dense kernels it replaces for long batches."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BETA = 100.0


def _weights(kin=72, hd=256, gf=256, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    W0 = torch.randn(hd, kin, generator=g) * (scale / kin ** 0.5)
    W0[:, kin - 1] = 0.0  # the pad column of the prepared weight
    b0 = torch.rand(hd, generator=g) * 0.2 - 0.1
    W1 = torch.randn(hd, hd, generator=g) * (scale / hd ** 0.5)
    b1 = torch.rand(hd, generator=g) * 0.2 - 0.1
    W2 = torch.randn(gf + 4, hd, generator=g) * (1.0 / hd ** 0.5)  # rows [feat | sdf | 0 0 0]
    b2 = torch.rand(gf + 4, generator=g) * 0.2 - 0.1
    return [t.to(DEV).requires_grad_(True) for t in (W0, b0, W1, b1, W2, b2)]


def _reference(E, ws, g_sdf):
    E64 = E.detach().double().cpu().requires_grad_(True)
    w64 = [w.detach().double().cpu().requires_grad_(True) for w in ws]
    W0, b0, W1, b1, W2, b2 = w64
    sp = lambda z: torch.nn.functional.softplus(z, beta=BETA)  # noqa: E731
    a1 = sp(sp(E64 @ W0.T + b0) @ W1.T + b1)
    gf = W2.shape[0] - 4
    sdf = a1 @ W2[gf] + b2[gf]
    grads = torch.autograd.grad((sdf * g_sdf.double().cpu()).sum(), [E64] + w64)
    return sdf.detach(), grads


def _run(E, ws, g_sdf, fused=True):
    """fused=True: the default policy (the sdf value chain kernels at any row count); False: NSKY_PRECISION=f32, the per-layer exact-fp32 path"""
    from neusky_amd import ops
    policy = ops._POLICY
    ops.set_precision_policy(policy if fused else "f32")
    try:
        ops.begin_step(DEV)
        Eg = E.clone().requires_grad_(True)
        sdf = ops.SDFValueFn.apply(Eg, *ws, BETA, True)
        grads = torch.autograd.grad((sdf * g_sdf).sum(), [Eg] + list(ws))
        torch.cuda.synchronize()
    finally:
        ops.set_precision_policy(policy)
    return sdf.detach(), grads


def _close(got, want, rel, what):
    want = want.to(torch.float64)
    err = (got.double().cpu() - want).abs().max().item()
    bar = rel * max(want.abs().max().item(), 1e-30)
    assert err <= bar, f"{what}: max err {err:.3e} > {bar:.3e}"


@pytest.mark.parametrize("M", [1, 77, 4096, 5000, 33000])
def test_fused_sdf_chain_matches_float64(M):
    g = torch.Generator().manual_seed(M)
    E = (torch.randn(M, 72, generator=g) * 0.5).to(DEV)
    E[:, 71] = 0.0
    ws = _weights(seed=M)
    g_sdf = torch.randn(M, generator=g).to(DEV)
    want_sdf, want = _reference(E, ws, g_sdf)
    sdf, got = _run(E, ws, g_sdf)
    _close(sdf, want_sdf, 2e-6, "sdf")
    names = ["dE", "dW0", "db0", "dW1", "db1", "dW2", "db2"]
    for n, a, b in zip(names, got, want):
        if n == "dE":
            a = a[:, :71]; b = b[:, :71]  # (the pad column multiplies a zero weight column in the product path; its gradient is unused)
        if n == "dW0":
            a = a[:, :71]; b = b[:, :71]
        _close(a, b, 3e-5, n)


def test_fused_sdf_chain_matches_per_layer_path_large_pre_activations():
    """beta z spans both softplus branches (linear above 20, exponential tail below): the saved-output sigmoid 1 - exp(-beta a)
    agrees with the per-layer kernels' separately saved sigmoid."""
    M = 8192 + 17
    g = torch.Generator().manual_seed(5)
    E = (torch.randn(M, 72, generator=g)).to(DEV)
    E[:, 71] = 0.0
    ws = _weights(seed=9, scale=2.0)
    g_sdf = torch.randn(M, generator=g).to(DEV)
    sdf_a, ga = _run(E, ws, g_sdf)
    sdf_b, gb = _run(E, ws, g_sdf, fused=False)
    want_sdf, want = _reference(E, ws, g_sdf)
    _close(sdf_a, want_sdf, 2e-6, "sdf fused")
    _close(sdf_b, want_sdf, 2e-6, "sdf per-layer")
    for n, a, b, w in zip(["dE", "dW0", "db0", "dW1", "db1", "dW2", "db2"], ga, gb, want):
        if n in ("dE", "dW0"):
            a, b, w = a[:, :71], b[:, :71], w[:, :71]
        _close(a, w, 3e-5, n + " fused")
        err_a = (a.double().cpu() - w).abs().max().item()
        err_b = (b.double().cpu() - w).abs().max().item()
        assert err_a <= max(4 * err_b, 1e-6 * w.abs().max().item()), f"{n}: fused {err_a:.3e} vs per-layer {err_b:.3e}"


def test_fused_sdf_chain_inference_rows_beyond_batch_untouched():
    """forward only, ragged batch: rows past M of the output buffer are not written"""
    from neusky_amd import hip
    M = 4096 + 40
    ws = [w.detach() for w in _weights(seed=3)]
    W0, b0, W1, b1, W2, b2 = ws
    gf = W2.shape[0] - 4
    net = hip.sdf_net(W0, b0, W1, b1, W2[gf], b2[gf:gf + 1], BETA)
    nbytes, ntiles = hip.sdf_stream_layout(net, 0)
    assert ntiles == 16
    stream = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    table = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
    hip.sdf_pack(net, stream, table, 0)
    E = torch.randn(M, 72, device=DEV) * 0.3
    A0 = torch.empty(hip.film_rows(M), 256, device=DEV); A1 = torch.empty_like(A0)
    out = torch.full((M + 64,), 7.0, device=DEV)
    hip.sdf_chain_fwd(net, stream, table, E, M, A0, A1, out)
    torch.cuda.synchronize()
    assert (out[M:] == 7.0).all() and torch.isfinite(out[:M]).all()
    sp = lambda z: torch.nn.functional.softplus(z, beta=BETA)  # noqa: E731
    want = sp(sp(E.double() @ W0.double().T + b0.double()) @ W1.double().T + b1.double()) @ W2[gf].double() + b2[gf].double()
    _close(out[:M], want.cpu(), 2e-6, "sdf")
