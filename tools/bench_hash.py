"""Micro-benchmark of the hash-grid encode kernels at the DDF / field sizes (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neusky_amd import hip
from neusky_amd.encoding import HashGridGeometry
dev = "cuda:0"
def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, P, smooth, mode, pe, tang, sphere in [("ddf cond", 262144, False, 0, 0, False, True), ("sdf probe", 262144, True, 1, 6, False, False),
                                                ("field +tangents", 98304, True, 1, 6, True, False)]:
    geom = HashGridGeometry(smoothstep=smooth)
    table = (torch.rand(geom.n_params, 2, device=dev) * 2 - 1) * 1e-2
    x = torch.rand(P, 3, device=dev) * 2 - 1
    if sphere: x = torch.nn.functional.normalize(x, dim=-1)
    else: x = x * 0.6
    width = 3 + 6 * pe + 32; ldy = (width + 3) // 4 * 4
    Y = torch.empty(P, ldy, device=dev); T = torch.empty(3, P, ldy, device=dev) if tang else None
    f = t(lambda: hip.encode_fwd(geom, table, x, mode, True, pe, 5.0, Y, T))
    dY = torch.randn(P, ldy, device=dev); dT = torch.randn(3, P, ldy, device=dev) if tang else None
    dtab = torch.zeros_like(table); dx = torch.empty(P, 3, device=dev)
    b = t(lambda: hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, dT, dtab, None))
    bx = t(lambda: hip.encode_bwd(geom, table, x, mode, True, pe, 5.0, dY, dT, dtab, dx))
    gb = P * 16 * 8 * 8 / 1e9
    print(f"{name}: P={P} fwd {f:.3f} ms ({gb/f*1e3:.0f} GB/s gathered)  bwd {b:.3f} ms ({P*256/b/1e6:.1f} G atomic floats/s)  bwd+dx {bx:.3f} ms")
