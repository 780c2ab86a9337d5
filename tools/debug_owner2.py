import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
from neusky_amd.encoding import HashGridGeometry
DEV = "cuda:0"
geom = HashGridGeometry(smoothstep=False)
g = torch.Generator().manual_seed(11)
table = ((torch.rand(geom.n_params, 2, generator=g) * 2 - 1) * 1e-2).to(DEV)
P = 40000
for name, fn in [("all positive", lambda x: x.abs()), ("x negative only", lambda x: torch.stack([-x[:, 0].abs(), x[:, 1].abs(), x[:, 2].abs()], 1)),
                 ("z negative only", lambda x: torch.stack([x[:, 0].abs(), x[:, 1].abs(), -x[:, 2].abs()], 1))]:
    x = torch.nn.functional.normalize(torch.rand(P, 3, generator=g) * 2 - 1, dim=-1)
    x = fn(x).to(DEV).contiguous()
    ldy = 36
    dY = torch.randn(P, ldy, generator=g).to(DEV)
    a = torch.zeros(geom.n_params, 2, device=DEV); b = torch.zeros_like(a)
    hip.encode_bwd(geom, table, x, 0, True, 0, 5.0, dY, None, a, None)
    hip.encode_bwd(geom, table, x, 0, True, 0, 5.0, dY, None, b, None, workspace=None)
    torch.cuda.synchronize()
    print(name)
    for lvl in (2, 3, 4):
        sl = slice(geom.offsets[lvl], geom.offsets[lvl + 1])
        d = (a[sl] - b[sl]).abs().sum(1)
        bad = torch.nonzero(d > 1e-3)[:, 0]
        print(f"  level {lvl} nbad {bad.numel()} first bad local idx {bad[:8].tolist()} a {a[sl][bad[:3]].tolist()} b {b[sl][bad[:3]].tolist()}")
