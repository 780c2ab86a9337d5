"""python tools/lab_film_fwd_bench.py <lib names or .so paths...>: forward kernel time of each lab library (subprocess each: NSKY_LIB is read at import)"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os, ctypes
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests"); sys.path.insert(0, "%s/tools")
import lab; lab.apply()
import torch
from neusky_amd import hip
from test_gpu_film_chain import _net, _inputs
from bench_film_kernels import NETS, timeit
DEV = "cuda:0"
for name in ("ddf", "illum"):
    H, n_map, n_film, cd, xd, od, M = NETS[name]
    net = _net(H, n_map, n_film, cd, xd, od)
    cond, x = _inputs(M, cd, xd); cond, x = cond.to(DEV), x.to(DEV)
    lins = net.mapping_network.linears()
    desc = hip.film_net(cd, xd, od, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                        [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
    nbytes, _ = hip.film_stream_layout(desc, 0)
    s = torch.zeros(nbytes, dtype=torch.uint8, device=DEV); t = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
    hip.film_pack(desc, s, t, 0)
    Mp = hip.film_rows(M)
    hs = [torch.empty(Mp, H, device=DEV) for _ in range(desc.n_map)]
    zs = [torch.empty(Mp, H, device=DEV) for _ in range(n_film)]
    ys = [torch.empty(Mp, H, device=DEV) for _ in range(n_film)]
    res = torch.empty(M, 4, device=DEV)
    tt = timeit(lambda: hip.film_chain_fwd(desc, s, t, cond, x, M, hs, zs, ys, res), 10)
    ys2 = [torch.empty(Mp, H, device=DEV) for _ in range(2)]
    ti = timeit(lambda: hip.film_chain_fwd(desc, s, t, cond, x, M, [hs[0]] * desc.n_map, None, [ys2[i %% 2] for i in range(n_film)], res), 10)
    print("  %%s: fwd %%7.1f us  (no saves %%7.1f us)" %% (name, tt * 1e3, ti * 1e3), flush=True)
    lib = ctypes.CDLL(hip.LIB_PATH)
    if hasattr(lib, "nsky_lab_stamps"):
        buf = (ctypes.c_ulonglong * 64)()
        lib.nsky_lab_stamps(buf)
        for role, off in (("A", 0), ("B", 16)):
            v = [buf[off + k] for k in range(16)]
            print("   stamps", role, [int(b - v[0]) for b in v[:6] if b], "slot20", [int(b - v[10]) for b in v[10:] if b], "MHz %%.0f" %% ((v[5] - v[0]) / max(1, v[9] - v[8]) * 100), flush=True)
''' % (R, R, R)
for n in sys.argv[1:]:
    lib = n if n.endswith(".so") else os.path.join(R, "scratch/r4/lab/libfwd_%s.so" % n)
    print(n, flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, NSKY_LIB=lib), timeout=300)
