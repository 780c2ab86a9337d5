import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
from neusky_amd.encoding import HashGridGeometry
DEV = "cuda:0"
for (smooth, mode, P) in [(False, 0, 263456), (False, 0, 40000), (True, 1, 98304)]:
    geom = HashGridGeometry(smoothstep=smooth)
    g = torch.Generator().manual_seed(11)
    table = ((torch.rand(geom.n_params, 2, generator=g) * 2 - 1) * 1e-2).to(DEV)
    x = torch.rand(P, 3, generator=g) * 2 - 1
    x = torch.nn.functional.normalize(x, dim=-1) if mode == 0 else x * 1.3
    x = x.to(DEV).contiguous()
    pe = 6 if mode == 1 else 0
    width = 3 + 6 * pe + 2 * geom.n_levels
    ldy = (width + 3) // 4 * 4
    dY = torch.randn(P, ldy, generator=g).to(DEV)
    a = torch.zeros(geom.n_params, 2, device=DEV); b = torch.zeros_like(a)
    hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, None, a, None)
    hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, None, b, None, workspace=None)
    torch.cuda.synchronize()
    print("case", smooth, mode, P)
    for lvl in range(geom.n_levels):
        sl = slice(geom.offsets[lvl], geom.offsets[lvl + 1])
        d = (a[sl] - b[sl]).abs()
        print(f"  level {lvl:2d} res {geom.resolutions[lvl]:5d} size {geom.offsets[lvl+1]-geom.offsets[lvl]:7d} max|b| {float(b[sl].abs().max()):9.4f} max err {float(d.max()):9.5f} sum a {float(a[sl].sum()):10.3f} sum b {float(b[sl].sum()):10.3f} nbad {int((d > 1e-3).sum())}")
