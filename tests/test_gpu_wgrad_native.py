"""nsky_wgrad_native (csrc/wgrad_native.hip): dW += dZ^T X and db += column sums of dZ over tile-native matrices, against
float64 matmuls of the same row-major data.  fp32-grade: the error bar is relative to |dZ|^T |X| (what an fp32 GEMM's own
rounding is measured against), not to the possibly cancelling result."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(M, n_out, k_in, seed, spread):
    g = torch.Generator().manual_seed(seed)
    dz = torch.randn(M, n_out, generator=g) * 1e-3
    if spread:  # gradients span many orders of magnitude across rows and features; a few rows dominate
        dz = dz * torch.exp(torch.randn(M, 1, generator=g) * 2.0) * torch.exp(torch.randn(1, n_out, generator=g))
    x = torch.sin(torch.randn(M, k_in, generator=g) * 3.0)
    if spread:
        x[:, : k_in // 4] *= 37.0  # leaky-ReLU style activations well above 1
        x[:, k_in // 4: k_in // 2] *= 1e-3
    return dz, x


@pytest.mark.parametrize("M,n_out,k_in,spread", [(4096, 256, 256, False), (5000, 256, 256, True), (33, 128, 128, True),
                                                 (70001, 2560, 256, True), (12345, 1280, 128, True), (1, 256, 128, False),
                                                 (263456, 256, 256, True)])
def test_wgrad_native_matches_float64(M, n_out, k_in, spread):
    from neusky_amd import hip
    dz, x = _case(M, n_out, k_in, 5, spread)
    dzd, xd = dz.to(DEV), x.to(DEV)
    A = hip.film_rows_to_native(dzd, n_out)
    B = hip.film_rows_to_native(xd, k_in)
    if M % 32:  # rows past M of the last block are never read as data: poison them
        hip.film_native_to_rows(A, hip.film_rows(M), n_out)  # (layout helper round trip stays usable on padded buffers)
        Ar = A.reshape(-1, n_out // 32, 4, 2, 32, 4)
        Ar[-1, :, :, :, M % 32:, :] = float("nan")
        Br = B.reshape(-1, k_in // 32, 4, 2, 32, 4)
        Br[-1, :, :, :, M % 32:, :] = float("nan")
    gmax = dzd.abs().max().reshape(1)
    dW0 = torch.randn(n_out, k_in, device=DEV)
    db0 = torch.randn(n_out, device=DEV)
    dW, db = dW0.clone(), db0.clone()
    hip.wgrad_native(A, n_out // 32, B, k_in // 32, M, dW, db, gmax)
    torch.cuda.synchronize()
    ref = dzd.double().T @ xd.double()
    bar = dzd.double().abs().T @ xd.double().abs()
    err = ((dW - dW0).double() - ref).abs()
    # fp32 accumulation of M products: ~sqrt(M) 2^-24 of the magnitude sum, the split adds ~2^-22 per product; the batch
    # splits are added one by one (fp32 atomics) onto what dW held: up to `splits` roundings at the running magnitude
    tol = 1e-6 * bar + 2e-5 * dW0.double().abs() + 1e-30
    assert bool((err <= tol).all()), f"dW: worst err/tol {float((err / tol).max()):.3e}"
    refb = dzd.double().sum(0)
    barb = dzd.double().abs().sum(0)
    errb = ((db - db0).double() - refb).abs()
    assert bool((errb <= 1e-6 * barb + 2e-5 * db0.double().abs()).all()), f"db: worst {float((errb / barb).max()):.3e}"


def test_wgrad_native_without_bias_and_scale_pointer():
    from neusky_amd import hip
    M, n_out, k_in = 2048, 128, 256
    dz, x = _case(M, n_out, k_in, 9, False)
    dz = dz * 1e3  # O(1) gradients: usable without the published maximum
    dzd, xd = dz.to(DEV), x.to(DEV)
    dW = torch.zeros(n_out, k_in, device=DEV)
    hip.wgrad_native(hip.film_rows_to_native(dzd, n_out), n_out // 32, hip.film_rows_to_native(xd, k_in), k_in // 32, M, dW)
    ref = dzd.double().T @ xd.double()
    bar = dzd.double().abs().T @ xd.double().abs()
    assert bool(((dW.double() - ref).abs() <= 4e-7 * bar).all())


@pytest.mark.parametrize("shapes", [[(256, 256), (2560, 256), (256, 256)], [(256, 256), (128, 256), (1280, 128)], [(256, 256)] * 16])
def test_wgrad_native_batch_is_the_sum_of_its_problems(shapes):
    """one launch for several layers sharing the batch (the FiLM chain's backward): every problem gets its own product"""
    from neusky_amd import hip
    M = 3000
    problems, checks = [], []
    for i, (n_out, k_in) in enumerate(shapes):
        dz, x = _case(M, n_out, k_in, 20 + i, True)
        dzd, xd = dz.to(DEV), x.to(DEV)
        A, B = hip.film_rows_to_native(dzd, n_out), hip.film_rows_to_native(xd, k_in)
        dW = torch.zeros(n_out, k_in, device=DEV)
        db = torch.zeros(n_out, device=DEV) if i % 2 == 0 else None
        gmax = dzd.abs().max().reshape(1)
        problems.append(hip.wgrad_problem(A, n_out // 32, B, k_in // 32, M, dW, db, gmax))
        checks.append((dzd, xd, dW, db, A, B, gmax))
    hip.wgrad_native_batch(problems, M)
    torch.cuda.synchronize()
    for dzd, xd, dW, db, *_ in checks:
        ref = dzd.double().T @ xd.double()
        bar = dzd.double().abs().T @ xd.double().abs()
        assert bool(((dW.double() - ref).abs() <= 1e-6 * bar).all())
        if db is not None:
            assert bool(((db.double() - dzd.double().sum(0)).abs() <= 1e-6 * dzd.double().abs().sum(0)).all())


@pytest.mark.parametrize("M,n_out,k_in,bias_rows", [(4096, 256, 256, 0), (393216, 256, 256, 98304), (98304, 260, 256, 0), (50001, 256, 72, 12500),
                                                     (33333, 256, 300, 0), (40000, 128, 64, 0)])
def test_wgrad_rowmajor_matches_float64(M, n_out, k_in, bias_rows):
    """row-major operands (the field's stacked value + tangent rows), 2-term bf16 products: 2^-16 per product relative to |dZ|^T |X|"""
    from neusky_amd import hip
    dz, x = _case(M, n_out, k_in, 7, True)
    lda, ldb = n_out + 4, k_in + 8  # leading dimensions wider than the matrices; the pad columns hold garbage
    A = torch.full((M, lda), float("nan")); A[:, :n_out] = dz
    B = torch.full((M, ldb), float("nan")); B[:, :k_in] = x
    Ad, Bd = A.to(DEV), B.to(DEV)
    dW0 = torch.randn(n_out, k_in + 4, device=DEV) * 1e-3
    db0 = torch.randn(n_out, device=DEV) * 1e-3
    dW, db = dW0.clone(), db0.clone()
    hip.wgrad_native_batch([hip.wgrad_problem_rowmajor(Ad[:, :n_out], n_out, Bd[:, :k_in], k_in, M, dW[:, :k_in], db, bias_rows)], M)
    torch.cuda.synchronize()
    dzd, xd = dz.to(DEV).double(), x.to(DEV).double()
    ref = dzd.T @ xd
    bar = dzd.abs().T @ xd.abs()
    err = ((dW - dW0)[:, :k_in].double() - ref).abs()
    assert bool((err <= 6e-5 * bar + 2e-5 * dW0[:, :k_in].double().abs() + 1e-30).all()), float((err / (bar + 1e-30)).max())
    assert torch.equal(dW[:, k_in:], dW0[:, k_in:]), "columns past k_in untouched"
    br = bias_rows if bias_rows else M
    refb = dzd[:br].sum(0)
    errb = ((db - db0).double() - refb).abs()
    assert bool((errb <= 2e-6 * dzd[:br].abs().sum(0) + 2e-5 * db0.double().abs()).all())


def test_grad_weight_routes_long_rowmajor_reductions_to_the_streaming_kernel():
    from neusky_amd import ops
    M, n_out, k_in = 40000, 256, 72
    dz, x = _case(M, n_out, k_in, 3, False)
    dzd, xd = dz.to(DEV), x.to(DEV)
    like, blike = torch.zeros(n_out, k_in, device=DEV), torch.zeros(n_out, device=DEV)
    dW, db = ops.grad_weight(dzd, xd, M, n_out, k_in, like, blike)
    ref = dzd.double().T @ xd.double()
    bar = dzd.double().abs().T @ xd.double().abs()
    assert bool(((dW.double() - ref).abs() <= 6e-5 * bar).all())
    assert torch.allclose(db.double(), dzd.double().sum(0), atol=1e-4, rtol=1e-4)


def test_wgrad_native_rejects_unsupported_widths():
    from neusky_amd import hip
    z = torch.zeros(32, 96, device=DEV)
    with pytest.raises(hip.NeuSkyHipError):
        hip.wgrad_native(z, 3, z, 3, 32, torch.zeros(96, 96, device=DEV))
