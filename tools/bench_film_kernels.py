"""standalone times of the three FiLM-SIREN chain kernels (forward with saves, FiLM backward, mapping backward) at the step's sizes,
one stream, HIP events: python tools/bench_film_kernels.py [ddf|illum] [reps]   (NSKY_LIB selects an experimental library build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
from neusky_amd import hip
from test_gpu_film_chain import _net, _inputs

DEV = "cuda:0"
NETS = {"ddf": (256, 5, 5, 35, 15, 1, 262144 + 1312), "illum": (128, 5, 9, 300, 10, 3, 153600 + 1024)}


def timeit(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    which = [a for a in sys.argv[1:] if a in NETS] or list(NETS)
    reps = next((int(a) for a in sys.argv[1:] if a.isdigit()), 10)
    for name in which:
        H, n_map, n_film, cd, xd, od, M = NETS[name]
        net = _net(H, n_map, n_film, cd, xd, od)
        cond, x = _inputs(M, cd, xd)
        cond, x = cond.to(DEV), x.to(DEV)
        lins = net.mapping_network.linears()
        desc = hip.film_net(cd, xd, od, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                            [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
        nm = desc.n_map
        packs = []
        for d in range(3):
            nbytes, _ = hip.film_stream_layout(desc, d)
            s = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
            t = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
            hip.film_pack(desc, s, t, d)
            packs.append((s, t))
        Mp = hip.film_rows(M)
        hs = [torch.empty(Mp, H, device=DEV) for _ in range(nm)]
        zs = [torch.empty(Mp, H, device=DEV) for _ in range(n_film)]
        ys = [torch.empty(Mp, H, device=DEV) for _ in range(n_film)]
        res = torch.empty(M, 4, device=DEV)
        t_fwd = timeit(lambda: hip.film_chain_fwd(desc, packs[0][0], packs[0][1], cond, x, M, hs, zs, ys, res), reps)
        d_res = torch.randn(M, 4, device=DEV)
        dzs = [torch.empty(Mp, H, device=DEV) for _ in range(n_film)]
        dfp = torch.empty(Mp, 2 * n_film * H, device=DEV)
        rowmax = torch.empty(Mp, device=DEV)
        gmax = torch.zeros(n_film + 1 + nm, device=DEV)
        d_x = torch.empty(M, x.stride(0), device=DEV)
        t_bf = timeit(lambda: hip.film_chain_bwd_film(desc, packs[1][0], packs[1][1], M, d_res, hs[-1], zs, dzs, dfp, rowmax, gmax[:n_film + 1], d_x), reps)
        dpres = [torch.empty(Mp, H, device=DEV) for _ in range(nm)]
        d_cond = torch.empty(M, cond.stride(0), device=DEV)
        t_bm = timeit(lambda: hip.film_chain_bwd_map(desc, packs[2][0], packs[2][1], M, dfp, rowmax, hs, dpres, d_cond, gmax[n_film + 1:]), reps)
        print(f"{name}: M={M} H={H}  fwd {t_fwd*1e3:7.1f} us   bwd_film {t_bf*1e3:7.1f} us   bwd_map {t_bm*1e3:7.1f} us   sum {(t_fwd+t_bf+t_bm)*1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
