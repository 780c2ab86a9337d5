"""Micro-benchmark of the streaming weight-gradient kernel at the DDF chain's sizes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
from neusky_amd import hip, ops
dev = "cuda:0"
M = 262144 + 1312
def t(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for n_out, k_in in [(256, 256), (2560, 256), (128, 128), (1280, 128)]:
    A = torch.randn(hip.film_rows(M), n_out, device=dev) * 1e-3
    B = torch.rand(hip.film_rows(M), k_in, device=dev)
    gmax = A.abs().max().reshape(1)
    dW = torch.zeros(n_out, k_in, device=dev); db = torch.zeros(n_out, device=dev)
    new = t(lambda: hip.wgrad_native(A, n_out // 32, B, k_in // 32, M, dW, db, gmax))
    like = torch.zeros(n_out, k_in, device=dev); bl = torch.zeros(n_out, device=dev)
    splits = ops._splits(M, n_out, k_in)
    old = t(lambda: hip.gemm(A, B, dW, n_out, k_in, M, a_kcontig=False, b_kcontig=False, k_splits=splits, precision=hip.PREC_F16X2,
                             a_native_nt=n_out // 32, b_native_nt=k_in // 32, a_scale_max=gmax, a_rowsum=db))
    byts = M * (n_out + k_in) * 4
    fl = 2.0 * M * n_out * k_in
    print(f"dW[{n_out},{k_in}] rows {M}: new {new:.3f} ms ({byts/new/1e6:.0f} GB/s, {fl/new/1e9:.0f} TF/s)   old {old:.3f} ms ({fl/old/1e9:.0f} TF/s)")

# the DDF chain's backward in one launch: 4 FiLM layers + mapping head + 2 mapping layers
probs, keep = [], []
for n_out, k_in in [(256, 256)] * 4 + [(2560, 256)] + [(256, 256)] * 2:
    A = torch.randn(hip.film_rows(M), n_out, device=dev) * 1e-3
    B = torch.rand(hip.film_rows(M), k_in, device=dev)
    gmax = A.abs().max().reshape(1)
    dW = torch.zeros(n_out, k_in, device=dev); db = torch.zeros(n_out, device=dev)
    keep.append((A, B, gmax, dW, db))
    probs.append(hip.wgrad_problem(A, n_out // 32, B, k_in // 32, M, dW, db, gmax))
ms = t(lambda: hip.wgrad_native_batch(probs, M))
byts = sum(M * 4 * (a.shape[1] + b.shape[1]) for a, b, *_ in keep)
fl = sum(2.0 * M * a.shape[1] * b.shape[1] for a, b, *_ in keep)
print(f"chain batch (7 problems, 16 blocks): {ms:.3f} ms ({byts/ms/1e6:.0f} GB/s of operands, {fl/ms/1e9:.0f} TF/s)")
