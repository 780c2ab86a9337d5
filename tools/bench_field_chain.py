"""time the four fused field kernels (+ their weight gradients) at the bench's point count: python tools/bench_field_chain.py [N] [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from neusky_amd import hip, ops
import test_gpu_field_chain as T
DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 99304
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ET = T._inputs(N, 4); ws = [w.detach() for w in T._weights(seed=5)]
W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ws
GF = 256
net = hip.field_net(72, 39, 100.0, b0, b1, W2[GF], b2[GF:GF + 1], b2[:GF], bc0, bc1, Wc2, bc2)
Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
e = lambda *s: torch.empty(*s, device=DEV)
a0q, a1q, Eq, a1max, sdf, grad = e(Mq, 256), e(Mq, 256), e(Mq, 128), e(N), e(N), e(N, 3)
a1v, feat, c0, c1, xpe, alb = e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 128), e(N, 4)
dpc2, dpc1, dpc0, dfeat, da1v, dxpe, gmax = e(N, 4), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(N, 40), torch.zeros(8, device=DEV)
d1q, d0q, dET = e(Mq, 256), e(Mq, 256), e(4 * N, 72)
g = torch.Generator().manual_seed(3)
g_sdf, g_grad, g_alb = torch.randn(N, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV), torch.randn(N, 3, generator=g).to(DEV)
pk1 = hip.chain_pack([hip.chain_layer(W0, 256, 72), hip.chain_layer(W1, 256, 256)], DEV)
pk2 = hip.chain_pack([hip.chain_layer(W2, 256, 256), hip.chain_layer(Wc0, 256, 300), hip.chain_layer(Wc1, 256, 256)], DEV)
pk3 = hip.chain_pack([hip.chain_layer(Wc1, 256, 256, True), hip.chain_layer(Wc0, 300, 256, True), hip.chain_layer(W2, 256, 256, True)], DEV)
pk4 = hip.chain_pack([hip.chain_layer(W1, 256, 256, True), hip.chain_layer(W0, 72, 256, True)], DEV)
dW = [torch.zeros_like(w) for w in ws]
R4 = 4 * N
steps = {
    "geo_fwd": lambda: hip.field_geo_fwd(net, pk1, ET, N, a0q, a1q, Eq, a1max, sdf, grad),
    "colour_fwd": lambda: hip.field_colour_fwd(net, pk2, ET, N, a1q, a1max, a1v, feat, xpe, c0, c1, alb),
    "colour_bwd": lambda: hip.field_colour_bwd(net, pk3, N, g_alb, alb, c0, c1, dpc2, dpc1, dpc0, dfeat, dxpe, da1v, gmax[:3]),
    "geo_bwd": lambda: hip.field_geo_bwd(net, pk4, N, g_sdf, g_grad, da1v, dxpe, a0q, a1q, d1q, d0q, dET, gmax[4:6]),
    "wgrad 4N wide": lambda: hip.wgrad_native_batch([hip.wgrad_problem(d1q, 8, a0q, 8, R4, dW[2], dW[3], gmax[4:5], 8.0, bias_row_mod=4)], R4),
    "wgrad 4N narrow": lambda: hip.wgrad_native_batch([hip.wgrad_problem(d0q, 8, Eq, 4, R4, dW[0], dW[1], gmax[5:6], 64.0, width_b=72, bias_row_mod=4)], R4),
    "colsum quad": lambda: hip.native_weighted_colsum(a1q, 8, R4, dW[4][GF], dW[5][GF:GF + 1], g_sdf=g_sdf, g_grad=g_grad),
    "wgrad N wide x3": lambda: hip.wgrad_native_batch([hip.wgrad_problem(dfeat, 8, a1v, 8, N, dW[4][:GF], dW[5][:GF], gmax[2:3], 8.0),
                                                        hip.wgrad_problem(dpc1, 8, c0, 8, N, dW[8], dW[9], gmax[0:1], 8.0),
                                                        hip.wgrad_problem(dpc0, 8, feat, 8, N, dW[6][:, :GF], dW[7], gmax[1:2], 8.0)], N),
    "wgrad N narrow": lambda: hip.wgrad_native_batch([hip.wgrad_problem(dpc0, 8, xpe, 4, N, dW[6][:, GF:], None, gmax[1:2], 64.0, width_b=44)], N),
    "colsum c1": lambda: hip.native_weighted_colsum(c1, 8, N, dW[10], dW[11], w4=dpc2, n_out=3),
}
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
tot = 0.0
for name, fn in steps.items():
    fn(); torch.cuda.synchronize()
    if only and name not in only:
        continue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tot += ms
    print(f"{name:18s} {ms * 1e3:8.1f} us")
print(f"total {tot:.3f} ms (one pass with colour; the step runs the geometry kernels twice)")
