"""The fused FiLM-SIREN chain kernel (nsky_film_pack + nsky_film_chain_fwd / _bwd) against a float64 restatement of
neusky/utils/siren.py:108-208 (oracle.film_siren, pinned by golden G8) and against the per-layer dense kernels."""
import pytest
import torch

from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(H, n_map, n_film, cond_dim, x_dim, out_dim, seed=0, device=DEV):
    from neusky_amd.utils.siren import FiLMSiren
    torch.manual_seed(seed)
    net = FiLMSiren(in_dim=x_dim, hidden_layers=n_film, hidden_features=H, mapping_network_in_dim=cond_dim,
                    mapping_network_layers=n_map, mapping_network_features=H, out_dim=out_dim).to(device)
    with torch.no_grad():  # biases away from zero so they are exercised
        for p in net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.3, 0.3)
    return net


def _oracle_params(net, dtype=torch.float64):
    c = lambda t: t.detach().cpu().to(dtype)  # noqa: E731
    p = {}
    lins = net.mapping_network.linears()
    for i, lin in enumerate(lins[:-1]):
        p[f"ddf.map_w{i}"], p[f"ddf.map_b{i}"] = c(lin.weight), c(lin.bias)
    p["ddf.map_wo"], p["ddf.map_bo"] = c(lins[-1].weight), c(lins[-1].bias)
    for i, l in enumerate(net.net):
        p[f"ddf.film_w{i}"], p[f"ddf.film_b{i}"] = c(l.layer.weight), c(l.layer.bias)
    p["ddf.out_w"], p["ddf.out_b"] = c(net.final_layer.weight), c(net.final_layer.bias)
    return p


def _inputs(M, cond_dim, x_dim, seed=1):
    g = torch.Generator().manual_seed(seed)
    pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
    cond = torch.zeros(M, pad4(cond_dim)); cond[:, :cond_dim] = torch.randn(M, cond_dim, generator=g) * 0.5
    x = torch.zeros(M, pad4(x_dim)); x[:, :x_dim] = torch.rand(M, x_dim, generator=g) * 2 - 1
    return cond, x


def _run_fused(net, cond, x, save=True):
    from neusky_amd import hip
    H, n_map, n_film = net.hidden, net.n_map, net.n_film
    lins = net.mapping_network.linears()
    desc = hip.film_net(net.cond_dim, net.in_dim, net.out_dim, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]],
                        lins[-1].weight, lins[-1].bias, [l.layer.weight for l in net.net], [l.layer.bias for l in net.net],
                        net.final_layer.weight, net.final_layer.bias)
    nbytes, ntiles = hip.film_stream_layout(desc)
    stream = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    scales = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
    hip.film_pack(desc, stream, scales)
    M = cond.shape[0]
    mk = lambda n: [torch.full((hip.film_rows(M), H), float("nan"), device=DEV) for _ in range(n)]  # noqa: E731
    hs, zs, ys = (mk(n_map), mk(n_film), mk(n_film)) if save else (None, None, mk(n_film))
    res = torch.full((M, 4), float("nan"), device=DEV)
    hip.film_chain_fwd(desc, stream, scales, cond.to(DEV), x.to(DEV), M, hs, zs, ys, res)
    torch.cuda.synchronize()
    rows = lambda ts: None if ts is None else [hip.film_native_to_rows(t, M, H) for t in ts]  # noqa: E731  saved in tile-native layout
    return res, rows(hs), rows(zs), rows(ys)


@pytest.mark.parametrize("H,n_map,n_film,cond_dim,x_dim,out_dim,M", [
    (256, 5, 5, 35, 15, 1, 300),      # the DDF network (neusky_config.py:163-177), ragged M
    (128, 5, 9, 300, 10, 3, 257),     # the RENI-shaped illumination decoder (latent 100 x 3)
    (128, 2, 3, 24, 10, 3, 64),       # small-latent test configuration
    (256, 1, 1, 16, 16, 4, 1),        # minimum depth, single row
])
def test_fused_forward_matches_float64_siren(H, n_map, n_film, cond_dim, x_dim, out_dim, M):
    net = _net(H, n_map, n_film, cond_dim, x_dim, out_dim)
    cond, x = _inputs(M, cond_dim, x_dim)
    res, hs, zs, ys = _run_fused(net, cond, x)
    p = _oracle_params(net)
    ref = O.film_siren(x[:, :x_dim].double(), cond[:, :cond_dim].double(), p)
    got = res[:, :out_dim].cpu().double()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    # float32 evaluation of the same chain (torch CPU) for scale: the SIREN frequencies ~30 amplify rounding
    p32 = _oracle_params(net, torch.float32)
    ref32 = O.film_siren(x[:, :x_dim].float(), cond[:, :cond_dim].float(), p32).double()
    err32 = (ref32 - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
    assert err < max(2e-5, 4 * err32), (err, err32)
    # saved activations: the last FiLM output and the last mapping activation against float64
    h = cond[:, :cond_dim].double()
    for i in range(n_map):
        h = torch.nn.functional.leaky_relu(torch.nn.functional.linear(h, p[f"ddf.map_w{i}"], p[f"ddf.map_b{i}"]), 0.2)
    assert (hs[-1].cpu().double() - h).abs().max().item() < 1e-5 * max(1.0, h.abs().max().item())
    assert all(torch.isfinite(t).all() for t in hs + zs + ys)
    # y = sin(freq z + phase) is consistent with the saved z of the same layer
    fo = torch.nn.functional.linear(h, p["ddf.map_wo"], p["ddf.map_bo"])
    i = n_film - 1
    freq, phase = fo[:, i * H:(i + 1) * H] * 15 + 30, fo[:, (n_film + i) * H:(n_film + i + 1) * H]
    y_from_z = torch.sin(freq * zs[i].cpu().double() + phase)
    assert (y_from_z - ys[i].cpu().double()).abs().max().item() < 2e-4


def test_fused_forward_without_saves_and_large_batch():
    """no-grad mode (render): only the hand-off buffers; M spans many workgroups; deterministic"""
    net = _net(256, 5, 5, 35, 15, 1)
    cond, x = _inputs(20000, 35, 15, seed=3)
    res_a, _, _, _ = _run_fused(net, cond, x, save=False)
    res_b, _, _, _ = _run_fused(net, cond, x, save=True)
    assert torch.equal(res_a[:, 0], res_b[:, 0])
    p = _oracle_params(net)
    sub = slice(19000, 20000)
    ref = O.film_siren(x[sub, :15].double(), cond[sub, :35].double(), p)[:, 0]
    assert (res_a[sub, 0].cpu().double() - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())


def _pack(desc, direction):
    from neusky_amd import hip
    nbytes, _ = hip.film_stream_layout(desc, direction)
    stream = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
    table = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
    hip.film_pack(desc, stream, table, direction)
    return stream, table


@pytest.mark.parametrize("H,n_map,n_film,cond_dim,x_dim,out_dim,M", [
    (256, 5, 5, 35, 15, 1, 300),
    (128, 5, 9, 300, 10, 3, 257),
    (128, 2, 3, 24, 10, 3, 64),
])
def test_fused_backward_matches_float64_autograd(H, n_map, n_film, cond_dim, x_dim, out_dim, M):
    """nsky_film_chain_bwd_film + _bwd_map against torch autograd of the float64 siren restatement: d_cond directly, and every
    parameter gradient formed (in float64, on the host) from the tile-native gradient matrices the kernels store"""
    from neusky_amd import hip
    net = _net(H, n_map, n_film, cond_dim, x_dim, out_dim)
    cond, x = _inputs(M, cond_dim, x_dim)
    lins = net.mapping_network.linears()
    desc = hip.film_net(cond_dim, x_dim, out_dim, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                        [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
    s0, t0 = _pack(desc, 0)
    s1, t1 = _pack(desc, 1)
    s2, t2 = _pack(desc, 2)
    Mp = hip.film_rows(M)
    mk = lambda n, w=H: [torch.full((Mp, w), float("nan"), device=DEV) for _ in range(n)]  # noqa: E731
    hs, zs, ys = mk(n_map), mk(n_film), mk(n_film)
    res = torch.empty(M, 4, device=DEV)
    hip.film_chain_fwd(desc, s0, t0, cond.to(DEV), x.to(DEV), M, hs, zs, ys, res)
    g = torch.Generator().manual_seed(5)
    d_res = torch.zeros(M, 4); d_res[:, :out_dim] = torch.randn(M, out_dim, generator=g)
    dzs, dpres = mk(n_film), mk(n_map)
    dfp = torch.full((Mp, 2 * n_film * H), float("nan"), device=DEV)
    rowmax = torch.full((Mp,), float("nan"), device=DEV)
    d_cond = torch.full((M, cond.shape[1]), float("nan"), device=DEV)
    gmax = torch.zeros(n_film + 1 + n_map, device=DEV)
    d_x = torch.full((M, x.shape[1]), float("nan"), device=DEV)
    hip.film_chain_bwd_film(desc, s1, t1, M, d_res.to(DEV), hs[-1], zs, dzs, dfp, rowmax, gmax[:n_film + 1], d_x)
    hip.film_chain_bwd_map(desc, s2, t2, M, dfp, rowmax, hs, dpres, d_cond, gmax[n_film + 1:])
    torch.cuda.synchronize()
    # ---- float64 reference
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(net).items()}
    c64 = cond[:, :cond_dim].double().requires_grad_(True)
    x64 = x[:, :x_dim].double().requires_grad_(True)
    out = O.film_siren(x64, c64, p)
    out.backward(d_res[:, :out_dim].double())
    rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)  # noqa: E731
    e = rel(d_cond[:, :cond_dim].cpu().double(), c64.grad)
    assert e < 2e-4, ("d_cond", e)
    assert rel(d_x[:, :x_dim].cpu().double(), x64.grad) < 2e-4 and float(d_x[:, x_dim:].abs().max() if d_x.shape[1] > x_dim else 0.0) == 0.0
    assert float(d_cond[:, cond_dim:].abs().max()) == 0.0 if d_cond.shape[1] > cond_dim else True
    rows = lambda t, w=H: hip.film_native_to_rows(t, M, w).cpu().double()  # noqa: E731
    y_prev = [x[:, :x_dim].double()] + [rows(t) for t in ys[:-1]]
    for i in range(n_film):
        dz = rows(dzs[i])
        assert rel(dz.t() @ y_prev[i], p[f"ddf.film_w{i}"].grad) < 3e-4, ("film_w", i)
        assert rel(dz.sum(0), p[f"ddf.film_b{i}"].grad) < 3e-4, ("film_b", i)
    h_prev = [cond[:, :cond_dim].double()] + [rows(t) for t in hs[:-1]]
    for l in range(n_map):
        dp = rows(dpres[l])
        assert rel(dp.t() @ h_prev[l], p[f"ddf.map_w{l}"].grad) < 3e-4, ("map_w", l)
        assert rel(dp.sum(0), p[f"ddf.map_b{l}"].grad) < 3e-4, ("map_b", l)
    dF = rows(dfp, 2 * n_film * H)
    assert rel(dF.t() @ rows(hs[-1]), p["ddf.map_wo"].grad) < 3e-4
    assert rel(dF.sum(0), p["ddf.map_bo"].grad) < 3e-4
    assert rel(rowmax[:M].cpu().double(), dF.abs().max(1).values) < 1e-6
    # published maxima of the gradient matrices (pre-scaling of the weight-gradient GEMMs)
    want = [rows(t).abs().max().item() for t in dzs] + [dF.abs().max().item()] + [rows(t).abs().max().item() for t in dpres]
    assert rel(gmax.cpu().double(), torch.tensor(want, dtype=torch.float64)) < 1e-6
    # ---- weight gradients by the GEMM kernel straight from the tile-native matrices (fp16 split pre-scaled by the published max)
    from neusky_amd import ops
    for i in range(1, n_film):
        like = net.net[i].layer.weight
        dW, db = ops.grad_weight(dzs[i], ys[i - 1], M, H, H, like, net.net[i].layer.bias, a_native_nt=H // 32, b_native_nt=H // 32, a_scale_max=gmax[i:i + 1])
        assert rel(dW.cpu().double(), p[f"ddf.film_w{i}"].grad) < 3e-4 and rel(db.cpu().double(), p[f"ddf.film_b{i}"].grad) < 3e-4, ("gemm film", i)
    lins_ = net.mapping_network.linears()
    dW, db = ops.grad_weight(dfp, hs[-1], M, 2 * n_film * H, H, lins_[-1].weight, lins_[-1].bias, a_native_nt=2 * n_film * H // 32, b_native_nt=H // 32,
                             a_scale_max=gmax[n_film:n_film + 1])
    assert rel(dW.cpu().double(), p["ddf.map_wo"].grad) < 3e-4 and rel(db.cpu().double(), p["ddf.map_bo"].grad) < 3e-4
    # first layers: native gradient against a row-major input (N <= 64: exact fp32 MFMA kernel)
    xp = x.to(DEV)
    w0 = ops.pad_weight(net.net[0].layer.weight)
    dW, db = ops.grad_weight(dzs[0], xp, M, H, xp.shape[1], w0, net.net[0].layer.bias, a_native_nt=H // 32)
    assert rel(dW[:, :x_dim].cpu().double(), p["ddf.film_w0"].grad) < 3e-4 and rel(db.cpu().double(), p["ddf.film_b0"].grad) < 3e-4


@pytest.mark.parametrize("H,n_map,n_film,cond_dim,x_dim,out_dim", [(256, 5, 5, 35, 15, 1), (128, 5, 9, 300, 10, 3)])
def test_fused_chain_is_bitwise_repeatable(H, n_map, n_film, cond_dim, x_dim, out_dim):
    """no atomics on the activation outputs of the three chain kernels: repeated launches agree to the bit (the rule
    tools/isa_lint.py checks statically; the mapping backward of round 3's build violated it)"""
    from neusky_amd import hip
    M = 66000 + 37
    net = _net(H, n_map, n_film, cond_dim, x_dim, out_dim)
    cond, x = _inputs(M, cond_dim, x_dim)
    lins = net.mapping_network.linears()
    desc = hip.film_net(cond_dim, x_dim, out_dim, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                        [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
    (s0, t0), (s1, t1), (s2, t2) = _pack(desc, 0), _pack(desc, 1), _pack(desc, 2)
    Mp = hip.film_rows(M)
    mk = lambda n, w=H: [torch.full((Mp, w), float("nan"), device=DEV) for _ in range(n)]  # noqa: E731
    g = torch.Generator().manual_seed(5)
    d_res = torch.zeros(M, 4); d_res[:, :out_dim] = torch.randn(M, out_dim, generator=g)
    d_res, cond, x = d_res.to(DEV), cond.to(DEV), x.to(DEV)
    first = None
    for it in range(16):
        hs, zs, ys, dzs, dpres = mk(n_map), mk(n_film), mk(n_film), mk(n_film), mk(n_map)
        res = torch.empty(M, 4, device=DEV)
        dfp = torch.full((Mp, 2 * n_film * H), float("nan"), device=DEV)
        rowmax = torch.full((Mp,), float("nan"), device=DEV)
        d_cond = torch.full((M, cond.shape[1]), float("nan"), device=DEV)
        d_x = torch.full((M, x.shape[1]), float("nan"), device=DEV)
        gmax = torch.zeros(n_film + 1 + n_map, device=DEV)
        hip.film_chain_fwd(desc, s0, t0, cond, x, M, hs, zs, ys, res)
        hip.film_chain_bwd_film(desc, s1, t1, M, d_res, hs[-1], zs, dzs, dfp, rowmax, gmax[:n_film + 1], d_x)
        hip.film_chain_bwd_map(desc, s2, t2, M, dfp, rowmax, hs, dpres, d_cond, gmax[n_film + 1:])
        torch.cuda.synchronize()
        cur = {"res": res[:, :out_dim], "z_last": zs[-1][:M], "dz0": dzs[0][:M], "dfp": dfp[:M], "dpre0": dpres[0][:M], "d_cond": d_cond[:, :cond_dim], "d_x": d_x[:, :x_dim]}
        if first is None:
            first = {k: v.clone() for k, v in cur.items()}
            continue
        for k, v in cur.items():
            bad = (v != first[k]).reshape(v.shape[0], -1).any(dim=1)
            assert not bool(bad.any()), f"launch {it}: {k} differs from launch 0 in {int(bad.sum())} rows, first {torch.nonzero(bad).flatten()[:8].tolist()}"
