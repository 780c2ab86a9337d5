"""time the fused FiLM-SIREN chain forward (and backward) against the per-layer dense-kernel path at the step's sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from neusky_amd import hip, ops
from test_gpu_film_chain import _net, _inputs

DEV = "cuda:0"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    for name, (H, n_map, n_film, cd, xd, od, M) in {"ddf": (256, 5, 5, 35, 15, 1, 262144 + 1312), "illum": (128, 5, 9, 300, 10, 3, 153600 + 1024)}.items():
        net = _net(H, n_map, n_film, cd, xd, od)
        cond, x = _inputs(M, cd, xd)
        cond, x = cond.to(DEV), x.to(DEV)
        lins = net.mapping_network.linears()
        desc = hip.film_net(cd, xd, od, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                            [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
        nbytes, ntiles = hip.film_stream_layout(desc)
        stream = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        scales = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
        hs = [torch.empty(hip.film_rows(M), H, device=DEV) for _ in range(n_map)]
        zs = [torch.empty(hip.film_rows(M), H, device=DEV) for _ in range(n_film)]
        ys = [torch.empty(hip.film_rows(M), H, device=DEV) for _ in range(n_film)]
        res = torch.empty(M, 4, device=DEV)
        t_pack = timeit(lambda: hip.film_pack(desc, stream, scales))
        t_save = timeit(lambda: hip.film_chain_fwd(desc, stream, scales, cond, x, M, hs, zs, ys, res))
        t_nosave = timeit(lambda: hip.film_chain_fwd(desc, stream, scales, cond, x, M, None, None, ys[:2] * n_film, res))
        flops = 2.0 * M * (cd * H + (n_map - 1) * H * H + H * 2 * n_film * H + xd * H + (n_film - 1) * H * H + H * od)
        wb = net.padded_weights()
        with torch.no_grad():
            t_old = timeit(lambda: ops.FilmSirenFn.apply(x, cond, n_map, n_film, False, False, *wb))
        print(f"{name}: M={M} stream {nbytes/1e6:.2f} MB, pack {t_pack*1e3:.1f} us, fused fwd (saving) {t_save:.3f} ms = {flops/t_save/1e9:.1f} TFLOP/s, "
              f"fused fwd (no saves) {t_nosave:.3f} ms = {flops/t_nosave/1e9:.1f} TFLOP/s, per-layer path (no grad) {t_old:.3f} ms")


if __name__ == "__main__":
    main()
