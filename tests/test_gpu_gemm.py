"""GPU parity of the fp32-MFMA dense layer (C ABI nsky_gemm_f32) against a float64 torch reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B, bias):
    y = A.double() @ B.double().T
    return y + bias.double() if bias is not None else y


def _mk(M, N, K, dev, seed=0, ldpad=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    A = torch.randn(M, K + ldpad, generator=g)[:, :K] if ldpad == 0 else torch.randn(M, K + ldpad, generator=g)
    return A


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 257, 72), (1000, 1, 256), (513, 3, 296), (2048, 256, 256),
                                   (77, 40, 16), (4096, 2560, 256)])
def test_forward_nt_bias_epilogues(M, N, K):
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / K**0.5
    b = torch.randn(N, device=dev)
    ldc = (N + 3) // 4 * 4
    Cbuf = torch.full((M, ldc), float("nan"), device=dev)
    hip.gemm(A, W, Cbuf, M, N, K, bias=b)
    ref = _ref(A, W, b)
    err = (Cbuf[:, :N].double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
    # fused activations
    for epi, fn in [(hip.EPI_RELU, torch.relu), (hip.EPI_LEAKY, lambda v: torch.nn.functional.leaky_relu(v, 0.2)),
                    (hip.EPI_SIGMOID, lambda v: 2.0 * torch.sigmoid(v)),
                    (hip.EPI_SOFTPLUS, lambda v: torch.nn.functional.softplus(v, beta=100))]:
        p0 = {hip.EPI_LEAKY: 0.2, hip.EPI_SIGMOID: 2.0, hip.EPI_SOFTPLUS: 100.0}.get(epi, 0.0)
        s1 = torch.empty(M, ldc, device=dev) if epi == hip.EPI_SOFTPLUS else None
        hip.gemm(A, W, Cbuf, M, N, K, bias=b, epi=epi, p0=p0, out1=s1)
        assert (Cbuf[:, :N].double() - fn(ref)).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
        if s1 is not None:
            assert (s1[:, :N].double() - torch.sigmoid(100 * ref)).abs().max().item() < 2e-3  # steep: 100x amplification


def test_film_epilogue_and_backward():
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(1)
    M, N, K = 700, 256, 256
    X = torch.rand(M, K, device=dev) * 2 - 1
    W = (torch.rand(N, K, device=dev) * 2 - 1) * (6 / K) ** 0.5 / 25
    b = torch.randn(N, device=dev) * 0.01
    F_ = torch.randn(M, N, device=dev) * 0.3
    P_ = torch.randn(M, N, device=dev)
    Y = torch.empty(M, N, device=dev)
    Z = torch.empty(M, N, device=dev)
    hip.gemm(X, W, Y, M, N, K, bias=b, epi=hip.EPI_FILM, p0=15.0, p1=30.0, aux0=F_, aux1=P_, out1=Z)
    z = _ref(X, W, b)
    y = torch.sin((15 * F_.double() + 30) * z + P_.double())
    assert (Z.double() - z).abs().max().item() < 1e-6
    assert (Y.double() - y).abs().max().item() < 5e-5
    # backward epilogue: dZ, dF, dP from an upstream GEMM result
    G = torch.randn(M, N, device=dev)
    eye = torch.eye(N, device=dev)
    dZ, dF, dP = (torch.empty(M, N, device=dev) for _ in range(3))
    hip.gemm(G, eye, dZ, M, N, N, epi=hip.EPI_BWD_FILM, p0=15.0, p1=30.0, aux0=Z, aux1=F_, aux2=P_, out1=dF, out2=dP)
    f = 15 * F_.double() + 30
    c = torch.cos(f * Z.double() + P_.double()) * G.double()
    assert (dZ.double() - c * f).abs().max().item() < 2e-3
    assert (dF.double() - c * Z.double() * 15).abs().max().item() < 2e-4
    assert (dP.double() - c).abs().max().item() < 2e-5


def test_backward_layouts_and_splitk():
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(2)
    M, N, K = 5000, 257, 72
    X = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    ldn = 260
    dYb = torch.zeros(M, ldn, device=dev)
    dYb[:, :N] = torch.randn(M, N, device=dev)
    dY = dYb[:, :N]
    # dX[M,K] = dY[M,N] @ W[N,K]      (A k-contig over N; B stored [Kred=N][K])
    dX = torch.empty(M, K, device=dev)
    # the reduction runs over the padded width ldn, so B carries zero rows up to ldn as well
    Wp = torch.zeros(ldn, K, device=dev); Wp[:N] = W
    hip.gemm(dYb, Wp, dX, M, K, ldn, a_kcontig=True, b_kcontig=False)
    ref = dY.double() @ W.double()
    assert (dX.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()
    # dW[N,K] = dY^T[N,M] @ X[M,K]    (both reduction-major) with split-K atomics
    dW = torch.zeros(N, K, device=dev)
    hip.gemm(dYb, X, dW, N, K, M, a_kcontig=False, b_kcontig=False, k_splits=16)
    refw = dY.double().T @ X.double()
    assert (dW.double() - refw).abs().max().item() < 1e-4 * refw.abs().max().item()
    db = torch.zeros(N, device=dev)
    hip.colsum(dYb, M, N, db)
    assert (db.double() - dY.double().sum(0)).abs().max().item() < 1e-3


def test_mul_aux_tangent_rows():
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(3)
    P, N, K = 333, 256, 72
    Xt = torch.randn(3 * P, K, device=dev)
    W = torch.randn(N, K, device=dev)
    S1 = torch.rand(P, N, device=dev)
    out = torch.empty(3 * P, N, device=dev)
    hip.gemm(Xt, W, out, 3 * P, N, K, epi=hip.EPI_MUL_AUX, aux0=S1, row_mod=P)
    ref = (Xt.double() @ W.double().T) * S1.double().repeat(3, 1)
    assert (out.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


def test_errors_are_reported():
    from neusky_amd import hip
    dev = "cuda:0"
    A = torch.zeros(8, 6, device=dev)
    with pytest.raises(hip.NeuSkyHipError):
        hip.gemm(A, A, torch.zeros(8, 8, device=dev), 8, 8, 6)  # lda % 4 != 0


@pytest.mark.parametrize("prec,tol", [(2, 6e-5), (3, 2e-6), (4, 2e-6)])
def test_split_bf16_precision_all_layouts(prec, tol):
    """operands split into 2 / 3 bf16 terms (3 / 6 bf16 MFMAs) or fp16 hi + scaled fp16 residual (3 fp16 MFMAs): every
    layout, ragged sizes, split-K + bias sums"""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(5)
    M, N, K = 1000, 260, 296
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    ref = _ref(A, W, b)
    sc = ref.abs().max().item()
    C = torch.full((M, N), float("nan"), device=dev)
    hip.gemm(A, W, C, M, N, K, bias=b, precision=prec)  # NT
    assert (C.double() - ref).abs().max().item() < tol * sc
    At = A.t().contiguous()  # [K, M]
    Wt = W.t().contiguous()  # [K, N]
    hip.gemm(A, Wt, C, M, N, K, bias=b, b_kcontig=False, precision=prec)  # NN
    assert (C.double() - ref).abs().max().item() < tol * sc
    hip.gemm(At, W, C, M, N, K, bias=b, a_kcontig=False, precision=prec)  # TN-ish (A reduction-major)
    assert (C.double() - ref).abs().max().item() < tol * sc
    C0 = torch.zeros(M, N, device=dev)
    rs = torch.zeros(M, device=dev)
    hip.gemm(At, Wt, C0, M, N, K, a_kcontig=False, b_kcontig=False, k_splits=4, a_rowsum=rs, precision=prec)
    assert (C0.double() - _ref(A, W, None)).abs().max().item() < tol * sc
    assert (rs.double() - A.double().sum(1)).abs().max().item() < 1e-3
    # fused backward epilogue rides on the same kernel
    aux = torch.randn(M, N, device=dev)
    hip.gemm(A, W, C, M, N, K, epi=hip.EPI_BWD_LEAKY, p0=0.2, aux0=aux, precision=prec)
    r2 = _ref(A, W, None)
    r2 = torch.where(aux.double() > 0, r2, 0.2 * r2)
    assert (C.double() - r2).abs().max().item() < tol * sc


def test_f16_scaled_split_range_and_accuracy():
    """NSKY_PREC_F16X2 (the default forward arithmetic): fp32-grade products (~2^-21 relative) for operands inside fp16's
    normal range (6.1e-5 <= |x| <= 65504); below it the representation error is an ABSOLUTE 2^-35 per operand (the
    residual is scaled by 2^11 before it is rounded), i.e. invisible next to the unit-scale terms every layer of this
    path also sums; magnitudes beyond 65504 saturate instead of turning into inf/NaN."""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(11)
    M, N, K = 512, 256, 256
    for sa, sw in ((1.0, 1.0), (1e-2, 1e-2), (2e3, 1e-3), (0.05, 30.0)):
        A = torch.randn(M, K, device=dev) * sa
        W = torch.randn(N, K, device=dev) * sw
        ref = A.double() @ W.double().T
        C = torch.empty(M, N, device=dev)
        hip.gemm(A, W, C, M, N, K, precision=hip.PREC_F16X2)
        err = (C.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-6, (sa, sw, err)
        hip.gemm(A, W, C, M, N, K, precision=hip.PREC_BF16X3)
        err3 = (C.double() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 4 * err3 + 2e-7, (sa, sw, err, err3)  # on a par with the 6-MFMA bf16 form
    # operands entirely below fp16's normal range: absolute floor 2^-35 per operand -> ~1e-5 relative at |a| ~ 3e-6
    A = torch.randn(M, K, device=dev) * 3e-6
    W = torch.randn(N, K, device=dev)
    ref = A.double() @ W.double().T
    C = torch.empty(M, N, device=dev)
    hip.gemm(A, W, C, M, N, K, precision=hip.PREC_F16X2)
    assert (C.double() - ref).abs().max().item() < 2.0 ** -35 * W.abs().max().item() * K
    # mixed magnitudes inside one row (hash features next to unit-scale encodings)
    A = torch.randn(M, K, device=dev) * torch.logspace(-6, 1, K, device=dev)
    W = torch.randn(N, K, device=dev)
    ref = A.double() @ W.double().T
    C = torch.empty(M, N, device=dev)
    hip.gemm(A, W, C, M, N, K, precision=hip.PREC_F16X2)
    assert (C.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    # out of range: saturates at +-65504, stays finite
    A = torch.full((M, K), 1e6, device=dev)
    W = torch.ones(N, K, device=dev) / K
    hip.gemm(A, W, C, M, N, K, precision=hip.PREC_F16X2)
    assert torch.isfinite(C).all() and abs(C[0, 0].item() - 65504.0) < 1.0


@pytest.mark.parametrize("M,N,K,prec_name", [(128, 128, 32, "f16x2"), (1000, 300, 256, "f16x2"), (4096, 2560, 256, "f16x2"),
                                             (777, 257, 64, "bf16x2"), (4099, 256, 2560, "bf16x2"), (2048, 128, 128, "f16x2"),
                                             (4096, 256, 72, "f16x2"), (1031, 256, 300, "bf16x2"), (513, 128, 36, "f16x2")])
def test_planes_kernel_matches_register_staged_kernel(M, N, K, prec_name):
    """nsky_split_planes + nsky_gemm_f32_planes (LDS-DMA kernel, pre-split weights): same MFMA sequence as nsky_gemm_f32 at
    that precision -> bit-identical results; and within the precision's bound of the float64 product.  Row tails (M % 128),
    column tails (N % 128), partial last k-tiles (K % 32), both plane orientations (forward layer / input gradient) and a
    fused epilogue."""
    from neusky_amd import hip
    dev = "cuda:0"
    prec = hip.PREC_F16X2 if prec_name == "f16x2" else hip.PREC_BF16X2
    torch.manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, device=dev)
    b = torch.randn(N, device=dev)
    ldc = (N + 3) // 4 * 4
    for transpose in (False, True):
        # the register-staged kernel wants ld % 4 == 0: a [K, N] weight lives in a padded buffer
        W = (torch.randn(K, ldc, device=dev) / K**0.5)[:, :N] if transpose else torch.randn(N, K, device=dev) / K**0.5
        planes = hip.split_planes(W, N, K, transpose, prec)
        assert planes.shape == (2, (N + 255) // 256 * 256, (K + 31) // 32 * 32) and planes.dtype == torch.int16
        ref_c = torch.full((M, ldc), float("nan"), device=dev)
        new_c = torch.full((M, ldc), float("nan"), device=dev)
        hip.gemm(A, W, ref_c, M, N, K, b_kcontig=not transpose, bias=b, epi=hip.EPI_LEAKY, p0=0.2, precision=prec)
        hip.gemm_planes(A, planes, new_c, M, N, K, precision=prec, bias=b, epi=hip.EPI_LEAKY, p0=0.2)
        assert torch.equal(ref_c[:, :N], new_c[:, :N])
        assert bool(torch.isnan(new_c[:, N:]).all())  # pad columns of C untouched
        ref = torch.nn.functional.leaky_relu(_ref(A, W.T if transpose else W, b), 0.2)
        tol = (3e-5 if prec == hip.PREC_F16X2 else 2e-4) * max(1.0, ref.abs().max().item())
        assert (new_c[:, :N].double() - ref).abs().max().item() < tol


def test_planes_kernel_film_backward_epilogue_and_argument_checks():
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(5)
    M, H = 1500, 256
    dZn = torch.randn(M, H, device=dev) * 1e-3
    W = torch.randn(H, H, device=dev) / 16
    z, F_, P_ = torch.randn(M, H, device=dev), torch.randn(M, H, device=dev) * 0.3, torch.randn(M, H, device=dev)
    outs = []
    for use_planes in (False, True):
        dz, dF, dP = (torch.empty(M, H, device=dev) for _ in range(3))
        kw = dict(epi=hip.EPI_BWD_FILM, p0=15.0, p1=30.0, aux0=z, aux1=F_, aux2=P_, out1=dF, out2=dP)
        if use_planes:
            hip.gemm_planes(dZn, hip.split_planes(W, H, H, True, hip.PREC_BF16X2), dz, M, H, H, precision=hip.PREC_BF16X2, **kw)
        else:
            hip.gemm(dZn, W, dz, M, H, H, a_kcontig=True, b_kcontig=False, precision=hip.PREC_BF16X2, **kw)
        outs.append((dz, dF, dP))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # K must be a multiple of 4 (16-byte chunks); the planes must carry the padding the kernel reads
    A = torch.randn(256, 52, device=dev)
    Wb = torch.randn(128, 52, device=dev)
    with pytest.raises(hip.NeuSkyHipError):
        hip.gemm_planes(A[:, :50], hip.split_planes(Wb, 128, 50, False, hip.PREC_F16X2), torch.empty(256, 128, device=dev), 256, 128, 50,
                        precision=hip.PREC_F16X2)
    with pytest.raises(hip.NeuSkyHipError):  # planes of a 32-wide split handed to a 52-deep product
        hip.gemm_planes(A, hip.split_planes(Wb, 128, 32, False, hip.PREC_F16X2), torch.empty(256, 128, device=dev), 256, 128, 52,
                        precision=hip.PREC_F16X2)
    with pytest.raises(hip.NeuSkyHipError):
        hip.gemm_planes(torch.randn(256, 64, device=dev), hip.split_planes(torch.randn(128, 64, device=dev), 128, 64, False, hip.PREC_F16X2),
                        torch.empty(256, 128, device=dev), 256, 128, 64, precision=hip.PREC_F32)


@pytest.mark.parametrize("M,N,ld", [(1000, 256, 256), (98304, 256, 260), (777, 37, 40), (4099, 4, 4), (513, 300, 300)])
def test_column_sums_plain_and_weighted(M, N, ld):
    """nsky_colsum_f32 / nsky_weighted_colsum_f32 (bias gradients, sdf-head weight gradient): float4 and scalar forms,
    accumulate semantics, strided weights."""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(M + N)
    X = torch.randn(M, ld, device=dev)[:, :N]
    out = torch.full((N,), 0.5, device=dev)
    hip.colsum(X, M, N, out)
    ref = 0.5 + X.double().sum(0)
    assert (out.double() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())
    W = torch.randn(M, 4, device=dev)
    out2 = torch.zeros(N, device=dev)
    hip.weighted_colsum(X, M, N, W[:, 1:2], 4, out2)  # weights read with stride 4
    ref2 = (W[:, 1:2].double() * X.double()).sum(0)
    assert (out2.double() - ref2).abs().max().item() < 1e-4 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("prec", ["f32", "bf16x2"])
def test_weight_gradient_row_sums_over_leading_rows_only(prec):
    """a_rowsum with rowsum_k_limit: dW = dZ^T X over all 4N stacked rows, db = column sums of dZ over the first N (value)
    rows only -- the bias gradient of the geo net's stacked [value; tangent] backward (sdf_albedo_field.py:235-238)."""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(11)
    N4, n_out, k_in, lim = 4 * 2048, 256, 72, 2048
    dZ = torch.randn(N4, n_out, device=dev)
    X = torch.randn(N4, k_in, device=dev)
    p = hip.PREC_F32 if prec == "f32" else hip.PREC_BF16X2
    for splits in (1, 8):
        dW = torch.zeros(n_out, k_in, device=dev)
        db = torch.zeros(n_out, device=dev)
        hip.gemm(dZ, X, dW, n_out, k_in, N4, a_kcontig=False, b_kcontig=False, k_splits=splits, a_rowsum=db, rowsum_k_limit=lim, precision=p)
        ref_w = dZ.double().T @ X.double()
        ref_b = dZ[:lim].double().sum(0)
        assert (dW.double() - ref_w).abs().max().item() < 2e-4 * ref_w.abs().max().item()
        assert (db.double() - ref_b).abs().max().item() < 1e-4 * ref_b.abs().max().item()
    with pytest.raises(hip.NeuSkyHipError):
        hip.gemm(dZ, X, dW, n_out, k_in, N4, a_kcontig=False, b_kcontig=False, k_splits=8, a_rowsum=db, rowsum_k_limit=100, precision=p)


def test_planes_kernel_race_screen():
    """The LDS-DMA kernel orders its DMA writes, in-place splits and fragment reads with hand-counted vmcnt waits and one
    barrier per k-tile: a mis-count shows up as rare wrong tiles that come and go with load.  Screen: 40 back-to-back launches
    at the step's largest shapes (other launches in flight, L2 / HBM busy), every result compared bit for bit with the
    register-staged kernel's."""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(9)
    for (M, N, K, prec, transpose) in [(262144, 256, 256, hip.PREC_F16X2, False), (65536, 256, 2560, hip.PREC_BF16X2, True),
                                       (153600, 128, 128, hip.PREC_F16X2, False), (98304, 256, 72, hip.PREC_F16X2, False)]:
        A = torch.randn(M, K, device=dev)
        W = (torch.randn(K, N, device=dev) if transpose else torch.randn(N, K, device=dev)) / K**0.5
        ref = torch.empty(M, N, device=dev)
        hip.gemm(A, W, ref, M, N, K, b_kcontig=not transpose, precision=prec)
        planes = hip.split_planes(W, N, K, transpose, prec)
        outs = [torch.empty(M, N, device=dev) for _ in range(4)]
        bad = 0
        for it in range(40):
            o = outs[it % 4]
            o.fill_(float("nan"))
            hip.gemm_planes(A, planes, o, M, N, K, precision=prec)
            if it % 4 == 3:
                bad += sum(int(not torch.equal(x, ref)) for x in outs)
        assert bad == 0, (M, N, K, bad)


def test_split_precisions_with_a_contraction_shorter_than_a_k_tile():
    """the input gradient of a narrow head (dX = dZ[M, 4] W[4, 128]) under the split policies: the split kernels stage whole 32-deep
    k-tiles, so a shorter contraction runs on the exact kernel (round 5: the 6-MFMA bf16 form on K = 4 read out of bounds)"""
    from neusky_amd import hip
    dev = "cuda:0"
    torch.manual_seed(5)
    for M, K in ((777, 4), (4096, 12), (64, 28)):
        dZ = torch.randn(M, K, device=dev)
        W = torch.randn(K, 128, device=dev)
        ref = dZ.double() @ W.double()
        for prec in (hip.PREC_BF16X3, hip.PREC_F16X2, hip.PREC_BF16X2):
            out = torch.full((M, 128), float("nan"), device=dev)
            hip.gemm(dZ, W, out, M, 128, K, a_kcontig=True, b_kcontig=False, precision=prec)
            assert (out.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item(), (M, K, prec)
