"""Parity at the BENCH workload's own size (VERDICT r2 "what's weak" 1 / "next round" 3): the kernels that dominate the step are
launched here with the grids, row counts and workgroup rounds of BASELINE configs[2] (1024 rays x 96 samples x 512 directions:
263 456 DDF rows = 1030 workgroups of the chain backward, 99 304 field points), and checked against the float64 oracle on slices
the oracle finishes in seconds -- every per-ray output depends on its own ray only."""
import os
import sys

import pytest
import torch

from oracle import neusky_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def full_forward():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from util_step import make_randoms, randomise, randoms_to
    torch.manual_seed(0)
    pipe = bench.build_pipeline(DEV, 1, 0)  # BASELINE configs[2] at full size
    randomise(pipe)
    R = bench.RAYS
    rb, batch = pipe.datamanager.next_train(0)
    rnd = make_randoms(pipe, R)
    pipe.model.set_step(10_000)
    outs, loss_dict, _ = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(rnd, DEV))
    torch.cuda.synchronize()
    return dict(pipe=pipe, rb=rb, batch=batch, rnd=rnd, outs=outs, loss_dict=loss_dict, R=R)


def test_full_size_forward_slice_against_the_oracle(full_forward):
    """16 rays of the 1024 x 96 x 512 training forward (spread over the batch, first and last ray included) through the float64
    oracle with their own jitters and camera rows: rendered radiance <= 1e-4 relative (north star), ray-sample indices bit-exact,
    DDF distances <= 3e-4, and the 1024-ray batch is finite everywhere"""
    from util_step import oracle_params, oracle_step_cfg
    f = full_forward
    pipe, rb, rnd, outs, R = f["pipe"], f["rb"], f["rnd"], f["outs"], f["R"]
    assert R == 1024 and pipe.model.config.num_neus_samples_per_ray == 96
    vd = outs["visibility_dict"]
    Dv = vd["visibility"].shape[1] // 2
    assert vd["visibility"].shape == (R, 512) and Dv == 256
    assert all(torch.isfinite(v).all() for v in (outs["rgb"], vd["visibility"], vd["expected_termination_dist"]))
    idx = torch.tensor([0, 1, 63, 64, 127, 255, 256, 300, 511, 512, 700, 767, 768, 900, 1022, 1023])
    p = oracle_params(pipe)
    cfg = oracle_step_cfg(pipe)
    light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).double()
    jit = [j[idx].double() for j in rnd["jitters"]]
    with torch.enable_grad():
        samp, fo, bg, p2p, vis, rgb = O.neusky_forward_rays(p, cfg, rb.origins.cpu().double()[idx], rb.directions.cpu().double()[idx],
                                                            rb.camera_indices.cpu().reshape(-1)[idx], jit, light)
    for got, ref in zip(outs["pdf_inds_list"], samp["inds_list"]):
        assert torch.equal(got.cpu().to(torch.int64)[idx], ref), "ray-sample (searchsorted) indices differ at the full size"
    got = outs["rgb"].detach().cpu().double()[idx]
    ref = rgb.detach()
    rel = (got - ref).abs().max() / ref.abs().max()
    assert rel < 1e-4, f"rendered radiance rel err {rel:.3e} at 1024 x 96 x 512"
    t_got = vd["expected_termination_dist"].detach().cpu().double().view(R, Dv)[idx]
    assert (t_got - vis["expected_termination_dist"].detach().view(len(idx), Dv)).abs().max() < 3e-4
    v_got = vd["visibility"].detach().cpu().double()[idx]
    assert (v_got - vis["visibility"].detach()).abs().max() < 6e-4
    assert (outs["p2p_dist"].detach().cpu().double()[idx] - p2p.detach()).abs().max() < 2e-4
    assert (outs["hdr_background_colours"].detach().cpu().double()[idx] - bg.detach()).abs().max() < 2e-4 * max(1.0, float(bg.abs().max()))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_full_size_slice.txt", "w") as fh:
        fh.write(f"1024x96x512 forward, 16-ray float64 oracle slice: rgb rel err {rel:.3e}; max |t - t_ref| "
                 f"{(t_got - vis['expected_termination_dist'].detach().view(len(idx), Dv)).abs().max():.3e}\n")


def test_full_size_chain_backward_rows_against_float64_autograd():
    """nsky_film_chain_fwd / _bwd_film / _bwd_map at the bench's M = 263 456 DDF rows (1030 workgroups of 256 rows: 4.02 rounds
    on 256 CUs, ragged last tile): d_cond, d_x of 512 sampled rows and the parameter-gradient contributions of those rows
    (dz_i^T y_(i-1), dpre_l^T h_(l-1), dfp^T h_last, formed in float64 from the tile-native matrices the kernels store) against
    torch autograd of the float64 siren restatement run on the same rows"""
    from neusky_amd import hip
    from test_gpu_film_chain import _inputs, _net, _oracle_params, _pack
    H, n_map, n_film, cond_dim, x_dim, out_dim = 256, 5, 5, 35, 15, 1
    M = 1024 * 256 + 1312
    net = _net(H, n_map, n_film, cond_dim, x_dim, out_dim)
    cond, x = _inputs(M, cond_dim, x_dim, seed=11)
    lins = net.mapping_network.linears()
    desc = hip.film_net(cond_dim, x_dim, out_dim, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                        [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
    s0, t0 = _pack(desc, 0)
    s1, t1 = _pack(desc, 1)
    s2, t2 = _pack(desc, 2)
    Mp = hip.film_rows(M)
    mk = lambda n, w=H: [torch.empty(Mp, w, device=DEV) for _ in range(n)]  # noqa: E731
    hs, zs, ys = mk(n_map), mk(n_film), mk(n_film)
    res = torch.empty(M, 4, device=DEV)
    cd, xd = cond.to(DEV), x.to(DEV)
    hip.film_chain_fwd(desc, s0, t0, cd, xd, M, hs, zs, ys, res)
    g = torch.Generator().manual_seed(5)
    d_res = torch.zeros(M, 4)
    d_res[:, :out_dim] = torch.randn(M, out_dim, generator=g)
    dzs, dpres = mk(n_film), mk(n_map)
    dfp = torch.empty(Mp, 2 * n_film * H, device=DEV)
    rowmax = torch.empty(Mp, device=DEV)
    d_cond = torch.full((M, cond.shape[1]), float("nan"), device=DEV)
    gmax = torch.zeros(n_film + 1 + n_map, device=DEV)
    d_x = torch.full((M, x.shape[1]), float("nan"), device=DEV)
    hip.film_chain_bwd_film(desc, s1, t1, M, d_res.to(DEV), hs[-1], zs, dzs, dfp, rowmax, gmax[:n_film + 1], d_x)
    hip.film_chain_bwd_map(desc, s2, t2, M, dfp, rowmax, hs, dpres, d_cond, gmax[n_film + 1:])
    torch.cuda.synchronize()
    assert torch.isfinite(d_cond).all() and torch.isfinite(d_x).all() and torch.isfinite(res[:, 0]).all()
    # 512 rows: the first and last workgroups, the ragged tail, and a spread over every round of the grid
    gi = torch.Generator().manual_seed(9)
    S = torch.cat([torch.arange(0, 40), torch.arange(M - 40, M), torch.randint(0, M, (432,), generator=gi)]).unique()
    Sd = S.to(DEV)
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(net).items()}
    c64 = cond[S, :cond_dim].double().requires_grad_(True)
    x64 = x[S, :x_dim].double().requires_grad_(True)
    out = O.film_siren(x64, c64, p)
    out.backward(d_res[S, :out_dim].double())
    rel = lambda a, b: (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)  # noqa: E731
    assert rel(res[Sd, :out_dim].cpu().double(), out.detach()) < 2e-5
    assert rel(d_cond[Sd, :cond_dim].cpu().double(), c64.grad) < 2e-4
    assert rel(d_x[Sd, :x_dim].cpu().double(), x64.grad) < 2e-4
    rows = lambda t, w=H: hip.film_native_to_rows(t, M, w)[Sd].cpu().double()  # noqa: E731
    y_prev = [x[S, :x_dim].double()] + [rows(t) for t in ys[:-1]]
    for i in range(n_film):
        dz = rows(dzs[i])
        assert rel(dz.t() @ y_prev[i], p[f"ddf.film_w{i}"].grad) < 3e-4, ("film_w", i)
        assert rel(dz.sum(0), p[f"ddf.film_b{i}"].grad) < 3e-4, ("film_b", i)
    h_prev = [cond[S, :cond_dim].double()] + [rows(t) for t in hs[:-1]]
    for l in range(n_map):
        dp = rows(dpres[l])
        assert rel(dp.t() @ h_prev[l], p[f"ddf.map_w{l}"].grad) < 3e-4, ("map_w", l)
        assert rel(dp.sum(0), p[f"ddf.map_b{l}"].grad) < 3e-4, ("map_b", l)
    dF = rows(dfp, 2 * n_film * H)
    assert rel(dF.t() @ rows(hs[-1]), p["ddf.map_wo"].grad) < 3e-4
    assert rel(dF.sum(0), p["ddf.map_bo"].grad) < 3e-4
