"""The fused SDF / albedo field kernels (csrc/field_chain.hip: nsky_field_geo_fwd / _colour_fwd / _colour_bwd / _geo_bwd,
ops.FieldChainFn) against a float64 torch restatement of SDFAlbedoField.get_outputs + get_colors
(neusky/fields/sdf_albedo_field.py:185-269) in which the input Jacobian rides in forward mode and torch autograd takes the
reverse of that (= the reference's double backward), and against the per-layer kernels they replace (ops.SDFAlbedoFn)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BETA = 100.0
KIN, HD, GF, LDC, NPE = 72, 256, 256, 300, 39


def _weights(seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    W0 = rn(HD, KIN) * (scale / KIN ** 0.5); W0[:, KIN - 1] = 0.0  # the pad column of the prepared weight
    W1 = rn(HD, HD) * (scale / HD ** 0.5)
    W2 = rn(GF + 4, HD) * (1.0 / HD ** 0.5); W2[GF + 1:] = 0.0     # rows [feat | sdf | 0 0 0]
    Wc0 = rn(HD, LDC) * (1.0 / LDC ** 0.5); Wc0[:, GF:GF + 4] = 0.0; Wc0[:, GF + 4 + NPE:] = 0.0  # [feat | 0 0 0 0 | x PE | 0]
    Wc1 = rn(HD, HD) * (1.4 / HD ** 0.5)
    Wc2 = rn(4, HD) * (1.0 / HD ** 0.5); Wc2[3] = 0.0
    b = lambda n: torch.rand(n, generator=g) * 0.2 - 0.1  # noqa: E731
    b2 = b(GF + 4); b2[GF + 1:] = 0.0
    bc2 = b(4); bc2[3] = 0.0
    ws = (W0, b(HD), W1, b(HD), W2, b2, Wc0, b(HD), Wc1, b(HD), Wc2, bc2)
    return [t.to(DEV).requires_grad_(True) for t in ws]


def _inputs(N, seed):
    g = torch.Generator().manual_seed(seed)
    ET = torch.zeros(4 * N, KIN)
    ET[:N, :KIN - 1] = torch.randn(N, KIN - 1, generator=g) * 0.5
    ET[N:, :KIN - 1] = torch.randn(3 * N, KIN - 1, generator=g) * 2.0  # tangent rows: larger, as PE derivatives are
    return ET.to(DEV)


def _gpu_relu_masks(ET, ws):
    """the ReLU decisions (c0 > 0, c1 > 0) of the fused forward, taken from its saved activations through the C ABI (the kernels are
    deterministic: the op makes the same ones).  A pre-activation within fp32 rounding of zero may legitimately land on the other
    side of the ReLU than in float64 (one in ~4e6 does at these sizes); the reference then differentiates the branch the kernel took."""
    from neusky_amd import hip
    W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = [w.detach() for w in ws]
    N = ET.shape[0] // 4
    net = hip.field_net(KIN, NPE, BETA, b0, b1, W2[GF], b2[GF:GF + 1], b2[:GF], bc0, bc1, Wc2, bc2)
    Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
    e = lambda *sh: torch.empty(*sh, device=DEV)  # noqa: E731
    a0q, a1q, a1max, sdf, grad = e(Mq, HD), e(Mq, HD), e(N), e(N), e(N, 3)
    hip.field_geo_fwd(net, hip.chain_pack([hip.chain_layer(W0, HD, KIN), hip.chain_layer(W1, HD, HD)], DEV), ET, N, a0q, a1q, None, a1max, sdf, grad)
    feat, c0, c1, xpe, alb = e(Mp, HD), e(Mp, HD), e(Mp, HD), e(Mp, 128), e(N, 4)
    hip.field_colour_fwd(net, hip.chain_pack([hip.chain_layer(W2, HD, HD), hip.chain_layer(Wc0, HD, LDC), hip.chain_layer(Wc1, HD, HD)], DEV), ET, N, a1q, a1max,
                         None, feat, xpe, c0, c1, alb)
    torch.cuda.synchronize()
    return (hip.film_native_to_rows(c0, N, HD) > 0).cpu(), (hip.film_native_to_rows(c1, N, HD) > 0).cpu()


def _reference(ET, ws, g_sdf, g_grad, g_alb, masks=None):
    N = ET.shape[0] // 4
    ET64 = ET.detach().double().cpu().requires_grad_(True)
    w64 = [w.detach().double().cpu().requires_grad_(True) for w in ws]
    W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = w64
    E, T = ET64[:N], ET64[N:].view(3, N, KIN)
    z0 = E @ W0.T + b0
    a0, s0 = F.softplus(z0, beta=BETA), torch.sigmoid(BETA * z0)
    ta0 = (T @ W0.T) * s0
    z1 = a0 @ W1.T + b1
    a1, s1 = F.softplus(z1, beta=BETA), torch.sigmoid(BETA * z1)
    ta1 = (ta0 @ W1.T) * s1
    sdf = a1 @ W2[GF] + b2[GF]
    grad = (ta1 @ W2[GF]).t()  # [N, 3]
    loss = (sdf * g_sdf.double().cpu()).sum() + (grad * g_grad.double().cpu()).sum()
    alb = None
    if g_alb is not None:
        feat = a1 @ W2[:GF].T + b2[:GF]
        cin = torch.cat([feat, torch.zeros(N, 4, dtype=torch.float64), E[:, :NPE], torch.zeros(N, LDC - GF - 4 - NPE, dtype=torch.float64)], 1)
        zc0 = cin @ Wc0.T + bc0
        if masks is not None:  # the kernel's ReLU decisions; they may differ from float64's only where the pre-activation is ~0
            for z, m in ((zc0, masks[0]),):
                flips = (z > 0) != m
                assert int(flips.sum()) <= 1 + z.numel() // 200000 and (not flips.any() or float(z[flips].abs().max()) < 2e-5 * float(z.abs().max()))
        c0 = torch.relu(zc0) if masks is None else zc0 * masks[0]
        zc1 = c0 @ Wc1.T + bc1
        if masks is not None:
            flips = (zc1 > 0) != masks[1]
            assert int(flips.sum()) <= 1 + zc1.numel() // 200000 and (not flips.any() or float(zc1[flips].abs().max()) < 2e-5 * float(zc1.abs().max()))
        c1 = torch.relu(zc1) if masks is None else zc1 * masks[1]
        alb = torch.sigmoid(c1 @ Wc2[:3].T + bc2[:3])
        loss = loss + (alb * g_alb.double().cpu()).sum()
    grads = torch.autograd.grad(loss, [ET64] + w64, allow_unused=True)
    return sdf.detach(), grad.detach(), None if alb is None else alb.detach(), grads


def _run(fn, ET, ws, g_sdf, g_grad, g_alb):
    from neusky_amd import ops
    ops.begin_step(DEV)
    Eg = ET.clone().requires_grad_(True)
    sdf, grad, alb = fn.apply(Eg, *ws, BETA, g_alb is not None)
    loss = (sdf * g_sdf).sum() + (grad * g_grad).sum()
    if g_alb is not None:
        loss = loss + (alb * g_alb).sum()
    grads = torch.autograd.grad(loss, [Eg] + list(ws), allow_unused=True)
    torch.cuda.synchronize()
    return sdf.detach(), grad.detach(), alb.detach(), grads


def _err(got, want):
    want = want.to(torch.float64)
    return (got.double().cpu() - want).abs().max().item(), max(want.abs().max().item(), 1e-30)


NAMES = ["dET", "dW0", "db0", "dW1", "db1", "dW2", "db2", "dWc0", "dbc0", "dWc1", "dbc1", "dWc2", "dbc2"]


def _crop(n, t):
    if n in ("dET", "dW0"):
        return t[:, :KIN - 1]  # the pad column multiplies a zero weight column / a zero input column: its gradient is unused
    if n == "dW2":
        return t[:GF + 1]
    if n == "db2":
        return t[:GF + 1]
    if n == "dWc2":
        return t[:3]
    if n == "dWc0":  # without the structural-zero columns (the sdf slot, the pads): their gradients are never read
        return torch.cat([t[:, :GF], t[:, GF + 4:GF + 4 + NPE]], 1)
    if n == "dbc2":
        return t[:3]
    return t


@pytest.mark.parametrize("N,colour", [(1024, True), (5003, True), (33001, True), (5003, False)])
def test_fused_field_matches_float64(N, colour):
    from neusky_amd import ops
    g = torch.Generator().manual_seed(N)
    ET = _inputs(N, N + 1)
    ws = _weights(seed=N)
    g_sdf, g_grad = torch.randn(N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
    g_alb = torch.randn(N, 3, generator=g).to(DEV) if colour else None
    want_sdf, want_grad, want_alb, want = _reference(ET, ws, g_sdf, g_grad, g_alb, _gpu_relu_masks(ET, ws) if colour else None)
    assert ops.field_fused_ok(ET, ws[0], ws[2], ws[4], ws[6], ws[8])
    sdf, grad, alb, got = _run(ops.FieldChainFn, ET, ws, g_sdf, g_grad, g_alb)
    for what, a, b in (("sdf", sdf, want_sdf), ("grad", grad, want_grad)) + ((("albedo", alb, want_alb),) if colour else ()):
        e, s = _err(a, b)
        assert e <= 3e-6 * s, f"{what}: {e:.3e} of {s:.3e}"
    if not colour:
        assert float(alb.abs().max()) == 0.0
    for n, a, b in zip(NAMES, got, want):
        if b is None or (not colour and n in NAMES[7:]):
            assert a is None or float(a.abs().max()) == 0.0, n
            continue
        if not colour and n in ("dW2", "db2"):  # only the sdf row trains
            e, s = _err(a[GF], b[GF])
        else:
            e, s = _err(_crop(n, a), _crop(n, b))
        assert e <= 4e-5 * s, f"{n}: max err {e:.3e} of {s:.3e}"


def test_fused_field_against_the_per_layer_path_both_softplus_branches():
    """beta z spans both softplus branches; the fused kernels (sigmoid recovered from the saved output, fp32-grade products both
    ways) are at least as close to float64 as the per-layer kernels they replace (2-term bf16 backward products)"""
    from neusky_amd import ops
    N = 8192 + 5
    g = torch.Generator().manual_seed(3)
    ET = _inputs(N, 4)
    ws = _weights(seed=5, scale=2.0)
    g_sdf, g_grad, g_alb = torch.randn(N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
    _, _, _, want = _reference(ET, ws, g_sdf, g_grad, g_alb, _gpu_relu_masks(ET, ws))
    _, _, _, want_b = _reference(ET, ws, g_sdf, g_grad, g_alb)  # (the per-layer kernels make their own ReLU decisions: float64's here)
    fa = _run(ops.FieldChainFn, ET, ws, g_sdf, g_grad, g_alb)
    fb = _run(ops.SDFAlbedoFn, ET, ws, g_sdf, g_grad, g_alb)
    for i, what in enumerate(("sdf", "grad", "albedo")):
        e, s = _err(fa[i], fb[i].double().cpu())
        assert e <= 1e-5 * s, (what, e, s)
    lines = []
    for n, a, b, w, wb in zip(NAMES, fa[3], fb[3], want, want_b):
        ea, s = _err(_crop(n, a), _crop(n, w))
        eb, _ = _err(_crop(n, b), _crop(n, wb))
        lines.append(f"{n:6s} fused {ea / s:.3e}  per-layer {eb / s:.3e}")
        assert ea <= 4e-5 * s, f"{n}: fused {ea:.3e} of {s:.3e}"
        assert ea <= max(4.0 * eb, 4e-6 * s), f"{n}: fused {ea:.3e} vs per-layer {eb:.3e}"
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    open("gpurun_out/r03_field_chain_grad_errors.txt", "w").write("gradient error / max|gradient| vs float64, N = 8197\n" + "\n".join(lines) + "\n")


def test_fused_field_inference_needs_no_saves_and_leaves_padding_alone():
    from neusky_amd import ops
    N = 4099
    ET = _inputs(N, 7)
    ws = [w.detach() for w in _weights(seed=8)]
    ops.begin_step(DEV)
    with torch.no_grad():
        sdf, grad, alb = ops.FieldChainFn.apply(ET, *ws, BETA, True)
    torch.cuda.synchronize()
    want_sdf, want_grad, want_alb, _ = _reference(ET, [w.clone().requires_grad_(True) for w in ws], torch.zeros(N, device=DEV), torch.zeros(N, 3, device=DEV),
                                                  torch.zeros(N, 3, device=DEV))
    assert sdf.shape == (N,) and grad.shape == (N, 3) and alb.shape == (N, 3)
    for a, b in ((sdf, want_sdf), (grad, want_grad), (alb, want_alb)):
        e, s = _err(a, b)
        assert e <= 3e-6 * s


def test_native_weighted_colsum_and_masked_bias():
    """the two helper paths of the field's parameter gradients on their own: weighted column sums over a tile-native matrix (row-major
    and quad weights) and the streaming weight-gradient kernel's value-rows-only bias sum / partial widths"""
    from neusky_amd import hip
    g = torch.Generator().manual_seed(1)
    N = 3001
    rows = 4 * N
    X = torch.randn(rows, 256, generator=g)
    Xn = hip.film_rows_to_native(X.to(DEV), 256)
    w4 = torch.randn(rows, 4, generator=g)
    out = torch.zeros(4, 256, device=DEV); bias = torch.zeros(4, device=DEV)
    hip.native_weighted_colsum(Xn, 8, rows, out, bias, w4=w4.to(DEV), n_out=3)
    want = w4[:, :3].double().t() @ X.double()
    assert (out[:3].cpu().double() - want).abs().max() <= 1e-4 * want.abs().max() and float(out[3].abs().max()) == 0.0
    assert (bias[:3].cpu().double() - w4[:, :3].double().sum(0)).abs().max() <= 1e-4 * w4.abs().sum(0).max()
    g_sdf, g_grad = torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
    out1 = torch.zeros(256, device=DEV); b1 = torch.zeros(1, device=DEV)
    hip.native_weighted_colsum(Xn, 8, rows, out1, b1, g_sdf=g_sdf.to(DEV), g_grad=g_grad.to(DEV))
    wq = torch.cat([g_sdf[:, None], g_grad], 1).reshape(-1).double()  # row 4 n + j
    want1 = wq @ X.double()
    assert (out1.cpu().double() - want1).abs().max() <= 1e-4 * want1.abs().max()
    assert abs(float(b1) - float(g_sdf.double().sum())) <= 1e-4 * float(g_sdf.abs().sum())
    # weight gradient with a quad-native dZ: bias over rows 4 n only, X with 72 of 128 columns
    dZ = torch.randn(rows, 256, generator=g) * 0.01
    Xe = torch.zeros(rows, 128); Xe[:, :72] = torch.randn(rows, 72, generator=g)
    dW = torch.zeros(256, 72, device=DEV); db = torch.zeros(256, device=DEV)
    gm = dZ.abs().max().reshape(1).to(DEV)
    hip.wgrad_native_batch([hip.wgrad_problem(hip.film_rows_to_native(dZ.to(DEV), 256), 8, hip.film_rows_to_native(Xe.to(DEV), 128), 4, rows, dW, db, gm, 64.0,
                                              width_b=72, bias_row_mod=4)], rows)
    wantW = dZ.double().t() @ Xe[:, :72].double()
    wantb = dZ.double()[0::4].sum(0)
    assert (dW.cpu().double() - wantW).abs().max() <= 3e-5 * wantW.abs().max()
    assert (db.cpu().double() - wantb).abs().max() <= 3e-5 * wantb.abs().max()


def test_fused_field_is_bitwise_repeatable():
    """The four field kernels have no atomics on their activation outputs: repeated launches on the same inputs must agree to the
    bit.  Round 3's build did not (tools/isa_lint.py, chain.h frag_settle): one wave's tile of d(encode rows) came out wrong in
    about one launch of six at this size, from weight fragments copied before their LDS read had landed."""
    from neusky_amd import ops
    N = 33001
    g = torch.Generator().manual_seed(N)
    ET = _inputs(N, N + 1)
    ws = _weights(seed=N)
    g_sdf, g_grad, g_alb = torch.randn(N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
    first = None
    for it in range(40):
        sdf, grad, alb, got = _run(ops.FieldChainFn, ET, ws, g_sdf, g_grad, g_alb)
        cur = {"sdf": sdf, "grad": grad, "albedo": alb, "dET": got[0][:, :KIN - 1]}
        if first is None:
            first = {k: v.clone() for k, v in cur.items()}
            continue
        for k, v in cur.items():
            bad = (v != first[k]).reshape(v.shape[0], -1).any(dim=1)
            assert not bool(bad.any()), f"launch {it}: {k} differs from launch 0 in {int(bad.sum())} rows, first {torch.nonzero(bad).flatten()[:8].tolist()}"


def test_fused_field_weight_gradients_with_steep_encode_tangents():
    """Late in training the finest hash levels are steep: d feature / d x reaches the thousands (a table difference of order one times a
    resolution of 2048).  The tangent rows of the encode matrix and of the hidden activations are operands of the first two layers'
    weight gradients; with a FIXED pre-scale of 64 their fp16 split overflowed (a 27 000-step run of the bench workload: NaN in
    d glin0 while every output and every other gradient was finite).  Operand scales now follow the published maxima."""
    from neusky_amd import ops
    N = 5003
    g = torch.Generator().manual_seed(N)
    ET = _inputs(N, N + 1)
    ET[N:] *= 4000.0 / float(ET[N:].abs().max())  # tangent rows up to 4000
    ws = _weights(seed=N, scale=0.05)  # (small first-layer weights keep the pre-activations in range)
    g_sdf, g_grad, g_alb = torch.randn(N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
    want_sdf, want_grad, want_alb, want = _reference(ET, ws, g_sdf, g_grad, g_alb, _gpu_relu_masks(ET, ws))
    sdf, grad, alb, got = _run(ops.FieldChainFn, ET, ws, g_sdf, g_grad, g_alb)
    for n, a, b in zip(NAMES, got, want):
        assert torch.isfinite(a).all(), n
        e, s = _err(_crop(n, a), _crop(n, b))
        assert e <= 1e-4 * s, f"{n}: max err {e:.3e} of {s:.3e}"
