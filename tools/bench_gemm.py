"""Micro-benchmark of nsky_gemm_f32 on the DDF layer shapes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip

dev = "cuda:0"
def run(M, N, K, epi=hip.EPI_NONE, iters=20, **kw):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K**0.5; b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev)
    extra = {}
    if epi == hip.EPI_FILM:
        extra = dict(aux0=torch.randn(M, N, device=dev), aux1=torch.randn(M, N, device=dev), p0=15.0, p1=30.0,
                     out1=torch.empty(M, N, device=dev))
    if epi == hip.EPI_LEAKY: extra = dict(p0=0.2)
    for _ in range(3): hip.gemm(A, W, C, M, N, K, bias=b, epi=epi, **extra, **kw)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hip.gemm(A, W, C, M, N, K, bias=b, epi=epi, **extra, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K} epi={epi}: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
    # compare to torch (rocBLAS/hipBLASLt) for context only
    for _ in range(3): torch.addmm(b, A, W.T)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): torch.addmm(b, A, W.T)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"   torch.addmm: {ms:.3f} ms  {2*M*N*K/ms/1e9:.1f} TFLOP/s")

run(262144, 256, 256)
run(262144, 256, 256, epi=hip.EPI_LEAKY)
run(262144, 256, 256, epi=hip.EPI_FILM)
run(262144, 2560, 256)
run(262144, 256, 36)
run(262144, 1, 256)
run(98304, 256, 72)
run(4096, 4096, 4096)

print("---- backward layouts at the DDF layer shape")
M, N, K = 262144, 256, 256
dZ = torch.randn(M, N, device=dev); W = torch.randn(N, K, device=dev); X = torch.randn(M, K, device=dev)
dX = torch.empty(M, K, device=dev); dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
def bench(fn, flops, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{name}: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s")
bench(lambda: hip.gemm(dZ, W, dX, M, K, N, a_kcontig=True, b_kcontig=False), 2.0*M*N*K, "dX = dZ @ W (NN)")
bench(lambda: hip.gemm(dZ, W, dX, M, K, N, a_kcontig=True, b_kcontig=False, epi=hip.EPI_BWD_LEAKY, p0=0.2, aux0=X), 2.0*M*N*K, "dX + BWD_LEAKY")
Z = torch.randn(M, N, device=dev); F_ = torch.randn(M, N, device=dev); P_ = torch.randn(M, N, device=dev); o1 = torch.empty(M, N, device=dev); o2 = torch.empty(M, N, device=dev)
bench(lambda: hip.gemm(dZ, W, dX, M, K, N, a_kcontig=True, b_kcontig=False, epi=hip.EPI_BWD_FILM, p0=15.0, p1=30.0, aux0=Z, aux1=F_, aux2=P_, out1=o1, out2=o2), 2.0*M*N*K, "dX + BWD_FILM")
for sp in (32, 64, 128, 256):
    bench(lambda: hip.gemm(dZ, X, dW, N, K, M, a_kcontig=False, b_kcontig=False, k_splits=sp, a_rowsum=db), 2.0*M*N*K, f"dW = dZ^T @ X (TN, splits={sp}, +rowsum)")
W2 = torch.randn(2560, 256, device=dev); dFP = torch.randn(M, 2560, device=dev); dh = torch.empty(M, 256, device=dev)
bench(lambda: hip.gemm(dFP, W2, dh, M, 256, 2560, a_kcontig=True, b_kcontig=False), 2.0*M*2560*256, "dh = dFP @ Wmo (NN, K=2560)")
dW2 = torch.zeros(2560, 256, device=dev)
bench(lambda: hip.gemm(dFP, X, dW2, 2560, 256, M, a_kcontig=False, b_kcontig=False, k_splits=64), 2.0*M*2560*256, "dWmo (TN, 2560x256, splits=64)")
