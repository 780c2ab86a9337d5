"""debug: colour-path intermediates of the fused field forward / backward against float64"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch, torch.nn.functional as F
from neusky_amd import hip, ops
import test_gpu_field_chain as T
DEV = "cuda:0"
N = 8197; scale = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
ET = T._inputs(N, 4); ws = [w.detach() for w in T._weights(seed=5, scale=scale)]
W0, b0, W1, b1, W2, b2, Wc0, bc0, Wc1, bc1, Wc2, bc2 = ws
GF = 256
net = hip.field_net(72, 39, 100.0, b0, b1, W2[GF], b2[GF:GF + 1], b2[:GF], bc0, bc1, Wc2, bc2)
Mq, Mp = hip.film_rows(4 * N), hip.film_rows(N)
e = lambda *s: torch.full(s, float("nan"), device=DEV)
a0q, a1q, a1max, sdf, grad = e(Mq, 256), e(Mq, 256), e(N), e(N), e(N, 3)
pk = hip.chain_pack([hip.chain_layer(W0, 256, 72), hip.chain_layer(W1, 256, 256)], DEV)
hip.field_geo_fwd(net, pk, ET, N, a0q, a1q, None, a1max, sdf, grad)
a1v, feat, c0, c1, xpe, alb = e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 128), e(N, 4)
pk2 = hip.chain_pack([hip.chain_layer(W2, 256, 256), hip.chain_layer(Wc0, 256, 300), hip.chain_layer(Wc1, 256, 256)], DEV)
hip.field_colour_fwd(net, pk2, ET, N, a1q, a1max, a1v, feat, xpe, c0, c1, alb)
torch.cuda.synchronize()
d = [w.double().cpu() for w in ws]
W0d, b0d, W1d, b1d, W2d, b2d, Wc0d, bc0d, Wc1d, bc1d, Wc2d, bc2d = d
E = ET.double().cpu()[:N]
a1 = F.softplus(F.softplus(E @ W0d.T + b0d, beta=100.) @ W1d.T + b1d, beta=100.)
featd = a1 @ W2d[:GF].T + b2d[:GF]
cin = torch.cat([featd, torch.zeros(N, 4, dtype=torch.float64), E[:, :39], torch.zeros(N, 1, dtype=torch.float64)], 1)
z0 = cin @ Wc0d.T + bc0d; c0d = torch.relu(z0); z1 = c0d @ Wc1d.T + bc1d; c1d = torch.relu(z1)
rows = lambda t, w=256: hip.film_native_to_rows(t, N, w).double().cpu()
r = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
print("a1v", r(rows(a1v), a1), "feat", r(rows(feat), featd), "max", featd.abs().max().item(), "xpe", r(rows(xpe, 128)[:, :44], cin[:, 256:300]))
g0 = rows(c0); g1 = rows(c1)
print("c0", r(g0, c0d), "abs", (g0 - c0d).abs().max().item(), "max", c0d.abs().max().item(), "flips", int(((g0 > 0) != (c0d > 0)).sum()), "min|z0| at flips",
      z0.abs()[(g0 > 0) != (c0d > 0)].max().item() if ((g0 > 0) != (c0d > 0)).any() else 0)
print("c1", r(g1, c1d), "flips", int(((g1 > 0) != (c1d > 0)).sum()))
# backward
g = torch.Generator().manual_seed(3)
g_alb = torch.randn(N, 3, generator=g).to(DEV)
dpc2, dpc1, dpc0, dfeat, da1v, dxpe, gmax = e(N, 4), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(Mp, 256), e(N, 40), torch.zeros(4, device=DEV)
pk3 = hip.chain_pack([hip.chain_layer(Wc1, 256, 256, True), hip.chain_layer(Wc0, 300, 256, True), hip.chain_layer(W2, 256, 256, True)], DEV)
hip.field_colour_bwd(net, pk3, N, g_alb, alb, c0, c1, dpc2, dpc1, dpc0, dfeat, dxpe, da1v, gmax[:3])
torch.cuda.synchronize()
albd = torch.sigmoid(c1d @ Wc2d[:3].T + bc2d[:3])
dp2 = g_alb.double().cpu() * albd * (1 - albd)
dp1 = (dp2 @ Wc2d[:3]) * (c1d > 0)
dp0 = (dp1 @ Wc1d) * (c0d > 0)
dcin = dp0 @ Wc0d
print("dpc2", r(dpc2.double().cpu()[:, :3], dp2), "dpc1", r(rows(dpc1), dp1), "dpc0", r(rows(dpc0), dp0), "dfeat", r(rows(dfeat), dcin[:, :256]),
      "dxpe", r(dxpe.double().cpu()[:, :39], dcin[:, 260:299]), "da1v", r(rows(da1v), dcin[:, :256] @ W2d[:GF]))
bad = (rows(dpc0) - dp0).abs()
i = bad.argmax(); print("worst dpc0 at", divmod(int(i), 256), "got", rows(dpc0).reshape(-1)[i].item(), "want", dp0.reshape(-1)[i].item(), "c0", g0.reshape(-1)[i].item(), c0d.reshape(-1)[i].item())
print("gmax", gmax.tolist(), [rows(dpc1).abs().max().item(), rows(dpc0).abs().max().item(), rows(dfeat).abs().max().item()])
