"""one full-size training step twice on identical inputs and injected randoms: fused FiLM chain vs the per-layer path; compares
loss terms and every optimizer group's gradient slab"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench
from neusky_amd import ops
from neusky_amd.engine import Optimizers, neusky_optimizers
from neusky_amd.model_components.losses import total_loss
from neusky_amd.utils.randomise import randomise
from util_step import make_randoms, randoms_to

dev = "cuda:0"
torch.manual_seed(1234)
R = int(os.environ.get("R", "1024"))
pipe = bench.build_pipeline(dev, 1, 0, rays=R)
randomise(pipe, seed=0)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
rb, batch = pipe.datamanager.next_train(0)
rnd = randoms_to(make_randoms(pipe, R), dev)
for k in ("light_rotation", "grid_perturb", "grid_dirs"):
    rnd[k] = rnd[k].to(dev)
res = {}
for name, thr in (("fused", 4096), ("layers", 1 << 30)):
    ops.FUSED_FILM_MIN_ROWS = thr
    opt.zero_grad_all()
    outs, ld, _ = pipe.get_train_loss_dict(2000, ray_bundle=rb, batch=batch, randoms=rnd)
    total_loss(ld).backward()
    torch.cuda.synchronize()
    res[name] = ({k: float(v.detach()) for k, v in ld.items()}, opt.flat_g.clone(), outs["rgb"].detach().clone())
    del outs, ld
a, b = res["fused"], res["layers"]
for k in a[0]:
    print(f"{k:34s} fused {a[0][k]:.6e} layers {b[0][k]:.6e}")
print("rgb max diff", float((a[2] - b[2]).abs().max()))
off = 0
for g in opt.groups:
    x, y = a[1][off:off + g.numel], b[1][off:off + g.numel]
    print(f"group {g.name:20s} max|g| {float(y.abs().max()):.3e} max diff {float((x - y).abs().max()):.3e} nan {bool(torch.isnan(x).any())}")
    # per parameter
    o2 = off
    for p in g.params:
        n = p.numel()
        xx, yy = a[1][o2:o2 + n], b[1][o2:o2 + n]
        d = float((xx - yy).abs().max()); m = float(yy.abs().max())
        if d > 1e-3 * m + 1e-12:
            print(f"     param shape {tuple(p.shape)} max {m:.3e} diff {d:.3e}")
        o2 += (n + 3) // 4 * 4
    off += g.numel
