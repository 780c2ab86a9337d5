"""graph replay vs eager gradients per parameter with the weight-gradient side stream on (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import torch
from test_gpu_graph import _setup
from neusky_amd.engine import GraphedTrainStep
from neusky_amd.model_components.losses import total_loss
import neusky_amd.ops as ops

pipe, opt, rb, batch, rnd = _setup()
step = 10_000
names = {p: n for n, p in pipe.named_parameters()}


def eager():
    opt.zero_grad_all()
    outs, ld, _ = pipe.get_train_loss_dict(step, ray_bundle=rb, batch=batch, randoms=rnd)
    total_loss(ld).backward()
    opt.collect_grads()
    torch.cuda.synchronize()
    return opt.flat_g.clone()


ops.ASYNC_WGRAD = False
g_sync = eager()
ops.ASYNC_WGRAD = True
g_async = eager()
g_async2 = eager()
stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=2, start_step=step, randoms=rnd)
stepper.load(rb, batch, sky=rnd["sky_ray_bundle"])
pipe.model.set_step(step)
stepper.graph.replay()
torch.cuda.synchronize()
g_graph = opt.flat_g.clone()
off = 0
for g in opt.groups:
    o = 0
    for p in g.params:
        k = p.numel()
        sl = slice(off + o, off + o + k)
        ref = g_sync[sl]
        sc = float(ref.abs().max()) + 1e-30
        e1, e2, e3 = (float((x[sl] - ref).abs().max()) / sc for x in (g_async, g_async2, g_graph))
        if max(e1, e2, e3) > 1e-5:
            print(f"{g.name:20s} {names[p]:60s} max|g| {sc:.3e}  async-eager {e1:.2e} {e2:.2e}  graph {e3:.2e}")
        o += (k + 3) // 4 * 4
    off += g.numel
print("done")
