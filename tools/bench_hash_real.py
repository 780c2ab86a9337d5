"""The table-gradient scatter (nsky_encode_bwd) on the point sets of a real train step: timing as they come (ray-major order) and with
the points randomly permuted (same work, no spatial coherence between neighbouring lanes).  Run on the GPU box."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import torch
import bench
from neusky_amd import hip
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from neusky_amd.utils.randomise import randomise

calls = []
orig = hip.encode_bwd


def spy(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, dY, dT, dtable, dx, **k):
    if x.shape[0] >= 90000:
        calls.append((geom, table, x.clone(), mode, include_x, pe_freqs, pe_max_exp, dY.clone(), None if dT is None else dT.clone(), dx is not None))
    return orig(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, dY, dT, dtable, dx, **k)


hip.encode_bwd = spy
pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
b = pipe.datamanager.next_train(0)
train_iteration(pipe, opt, 1000, ray_bundle=b[0], batch=b[1])
torch.cuda.synchronize()
hip.encode_bwd = orig


def t(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for geom, table, x, mode, include_x, pe, pmax, dY, dT, want_dx in calls:
    P = x.shape[0]
    dtab = torch.zeros_like(table)
    a = t(lambda: orig(geom, table, x, mode, include_x, pe, pmax, dY, dT, dtab, None))
    perm = torch.randperm(P, device=x.device)
    xp, dYp = x[perm].contiguous(), dY[perm].contiguous()
    dTp = None if dT is None else dT[:, perm].contiguous()
    b_ = t(lambda: orig(geom, table, xp, mode, include_x, pe, pmax, dYp, dTp, dtab, None))
    xr = (torch.rand_like(x) * 2 - 1) * 0.6
    c = t(lambda: orig(geom, table, xr, mode, include_x, pe, pmax, dY, dT, dtab, None))
    ext = x.abs().amax(0).tolist()
    print(f"P={P} levels={geom.n_levels} T={tuple(table.shape)} mode={mode} tangents={dT is not None}: as-is {a:.3f} ms  permuted {b_:.3f} ms  uniform-random x {c:.3f} ms  |x|max {ext}")
    # the dense levels alone (the first levels of the same geometry) and how concentrated the points are
    import copy
    nd = sum(1 for l in range(geom.n_levels) if geom.offsets[l + 1] - geom.offsets[l] < (1 << geom.log2_hashmap_size))
    if 0 < nd < geom.n_levels:
        sub = copy.copy(geom)
        sub.n_levels, sub.scales, sub.resolutions, sub.offsets = nd, geom.scales[:nd], geom.resolutions[:nd], geom.offsets[:nd + 1]
        tb, dtb = table[:sub.n_params].contiguous(), torch.zeros(sub.n_params, 2, device=x.device)
        d = t(lambda: orig(sub, tb, x, mode, include_x, pe, pmax, dY, dT, dtb, None))
        e = t(lambda: orig(sub, tb, xr, mode, include_x, pe, pmax, dY, dT, dtb, None))
        cells = []
        for l in (0, nd - 1, min(nd + 3, geom.n_levels - 1)):
            q = ((x.clamp(-1, 1) * 0.5 + 0.5) * geom.scales[l] + 0.5).floor().long()
            cells.append((geom.resolutions[l], int(torch.unique(q[:, 0] * 4096 * 4096 + q[:, 1] * 4096 + q[:, 2]).numel())))
        per = []
        for k in range(1, geom.n_levels + 1):
            sk = copy.copy(geom)
            sk.n_levels, sk.scales, sk.resolutions, sk.offsets = k, geom.scales[:k], geom.resolutions[:k], geom.offsets[:k + 1]
            tk, dk = table[:sk.n_params].contiguous(), torch.zeros(sk.n_params, 2, device=x.device)
            per.append(round(t(lambda: orig(sk, tk, x, mode, include_x, pe, pmax, dY, dT, dk, None)), 3))
        print(f"    first k levels, k = 1..{geom.n_levels}: {per} ms")
        print(f"    dense levels only ({nd}): as-is {d:.3f} ms, uniform-random {e:.3f} ms; distinct cells (res, n): {cells}")
