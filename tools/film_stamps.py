"""diagnostic build only (FILM_LAB_STAMP): where one workgroup of the fused FiLM-SIREN forward spends its cycles"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from neusky_amd import hip
from test_gpu_film_chain import _net, _inputs
DEV = "cuda:0"
lib = hip._lib
H, n_map, n_film, cd, xd, od, M = 256, 5, 5, 35, 15, 1, 262144 + 1312
net = _net(H, n_map, n_film, cd, xd, od)
cond, x = _inputs(M, cd, xd)
cond, x = cond.to(DEV), x.to(DEV)
lins = net.mapping_network.linears()
desc = hip.film_net(cd, xd, od, [l.weight for l in lins[:-1]], [l.bias for l in lins[:-1]], lins[-1].weight, lins[-1].bias,
                    [l.layer.weight for l in net.net], [l.layer.bias for l in net.net], net.final_layer.weight, net.final_layer.bias)
nbytes, ntiles = hip.film_stream_layout(desc)
stream = torch.zeros(nbytes, dtype=torch.uint8, device=DEV); scales = torch.empty(hip.FILM_TABLE_FLOATS, device=DEV)
hip.film_pack(desc, stream, scales)
hs = [torch.empty(M, H, device=DEV) for _ in range(n_map)]; zs = [torch.empty(M, H, device=DEV) for _ in range(n_film)]
ys = [torch.empty(M, H, device=DEV) for _ in range(n_film)]; res = torch.empty(M, 4, device=DEV)
for _ in range(3):
    hip.film_chain_fwd(desc, stream, scales, cond, x, M, hs, zs, ys, res)
torch.cuda.synchronize()
lib.nsky_film_lab_stamps(None, 1)
hip.film_chain_fwd(desc, stream, scales, cond, x, M, hs, zs, ys, res)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
lib.nsky_film_lab_stamps(out, 0)
names = ["product F (40 tiles x 16 k-steps)", "products P + Z", "tile epilogue VALU", "tile stores", "hand-off read-back (5 layers)", "whole FiLM phase", "mapping phase"]
for i, n in enumerate(names):
    print(f"{n:40s} {out[i]:10d} ticks (100 MHz) = {out[i]*10/1e3:8.1f} us")
