"""debug: intermediates of the fused field forward against float64 (quad-native saves, per row-set)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, torch.nn.functional as F
from neusky_amd import hip, ops
import test_gpu_field_chain as T
DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ET = T._inputs(N, 1); ws = [w.detach() for w in T._weights(seed=2)]
W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ws
GF = 256
net = hip.field_net(72, 39, 100.0, b0, b1, W2[GF], b2[GF:GF + 1], b2[:GF], bc0, bc1, Wc2, bc2)
Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
a0q, a1q = torch.full((Mq, 256), float("nan"), device=DEV), torch.full((Mq, 256), float("nan"), device=DEV)
Eq = torch.full((Mq, 128), float("nan"), device=DEV); a1max = torch.full((N,), float("nan"), device=DEV)
sdf, grad = torch.full((N,), float("nan"), device=DEV), torch.full((N, 3), float("nan"), device=DEV)
pk = hip.chain_pack([hip.chain_layer(W0, 256, 72), hip.chain_layer(W1, 256, 256)], DEV)
hip.field_geo_fwd(net, pk, ET, N, a0q, a1q, Eq, a1max, sdf, grad)
torch.cuda.synchronize()
E64 = ET.double().cpu(); w64 = [w.double().cpu() for w in ws]
W0d, b0d, W1d, b1d, W2d, b2d = w64[:6]
E, Tt = E64[:N], E64[N:].view(3, N, 72)
z0 = E @ W0d.T + b0d; a0 = F.softplus(z0, beta=100.); s0 = torch.sigmoid(100. * z0); ta0 = (Tt @ W0d.T) * s0
z1 = a0 @ W1d.T + b1d; a1 = F.softplus(z1, beta=100.); s1 = torch.sigmoid(100. * z1); ta1 = (ta0 @ W1d.T) * s1
r = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
g0 = hip.quad_native_to_rows(a0q, N, 256); g1 = hip.quad_native_to_rows(a1q, N, 256); ge = hip.quad_native_to_rows(Eq, N, 128)
print("Eq value", r(ge[0][:, :72], E), "Eq tangents", [r(ge[k + 1][:, :72], Tt[k]) for k in range(3)], "pad", float(ge[:, :, 72:].abs().max()))
print("a0 value", r(g0[0], a0), "ta0", [r(g0[k + 1], ta0[k]) for k in range(3)])
print("a1 value", r(g1[0], a1), "ta1", [r(g1[k + 1], ta1[k]) for k in range(3)])
print("sdf", r(sdf, a1 @ W2d[GF] + b2d[GF]), "grad", r(grad, (ta1 @ W2d[GF]).t()), "a1max", r(a1max, a1.abs().max(1).values))
print("g0 tangent sample", g0[1][0, :6].tolist(), ta0[0][0, :6].tolist())
