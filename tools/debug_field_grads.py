"""debug: gradient errors of the fused field vs per-layer path vs float64, per tensor (and per row-set for dET)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from neusky_amd import ops
import test_gpu_field_chain as T
N = 8192 + 5
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
g = torch.Generator().manual_seed(3)
ET = T._inputs(N, 4); ws = T._weights(seed=5, scale=scale)
g_sdf, g_grad, g_alb = torch.randn(N, generator=g).to(T.DEV), torch.randn(N, 3, generator=g).to(T.DEV), torch.randn(N, 3, generator=g).to(T.DEV)
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
if mode == "sdf":
    g_grad = g_grad * 0; g_alb = g_alb * 0
if mode == "grad":
    g_sdf = g_sdf * 0; g_alb = g_alb * 0
if mode == "alb":
    g_sdf = g_sdf * 0; g_grad = g_grad * 0
_, _, _, want = T._reference(ET, ws, g_sdf, g_grad, g_alb)
fa = T._run(ops.FieldChainFn, ET, ws, g_sdf, g_grad, g_alb)
fb = T._run(ops.SDFAlbedoFn, ET, ws, g_sdf, g_grad, g_alb)
for n, a, b, w in zip(T.NAMES, fa[3], fb[3], want):
    ea, s = T._err(T._crop(n, a), T._crop(n, w)); eb, _ = T._err(T._crop(n, b), T._crop(n, w))
    print(f"{n:6s} fused {ea / s:.3e}  per-layer {eb / s:.3e}   max {s:.3e}")
a, b, w = fa[3][0], fb[3][0], want[0]
for j in range(4):
    sl = slice(j * N, (j + 1) * N)
    for lo, hi, nm in ((0, 39, "xpe"), (39, 71, "hash")):
        ea, s = T._err(a[sl, lo:hi], w[sl, lo:hi]); eb, _ = T._err(b[sl, lo:hi], w[sl, lo:hi])
        print(f"dET row-set {j} {nm}: fused {ea:.3e} per-layer {eb:.3e} max {s:.3e}")
