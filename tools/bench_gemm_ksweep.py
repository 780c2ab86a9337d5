"""fixed per-workgroup overhead of the GEMM kernels: time vs K at fixed M, N (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
dev = "cuda:0"
def run(M, N, K, prec, iters=20, **kw):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / 16; C = torch.empty(M, N, device=dev)
    fn = lambda: hip.gemm(A, W, C, M, N, K, precision=prec, **kw)
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for prec in (3, 2, 0):
    for N in (256, 2560):
        M = 262144 if N == 256 else 65536
        ts = [(K, run(M, N, K, prec)) for K in (64, 128, 256, 512, 1024)]
        print(f"prec {prec} M={M} N={N}: " + " | ".join(f"K={K}: {t*1e3:.0f}us {2*M*N*K/t/1e9:.0f}TF" for K, t in ts))
