"""python tools/flake_graph.py [pipelines] [replays] : tests/test_gpu_graph.py::test_graph_replay_equals_eager_on_injected_randoms in a loop
(it failed once in a full -m gpu run: one 'fields' gradient 1e-3 of the group's max away from the eager step's).  Several eager passes and
several replays of each captured step; every mismatch is reported with the parameter it falls in -- which side moves, eager or replay?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p_)
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
import test_gpu_graph as T
from neusky_amd.engine import GraphedTrainStep
from neusky_amd.model_components.losses import total_loss

n_pipes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def where(pipe, opt, idx):
    names = {id(p): n for n, p in pipe.named_parameters()}
    off = 0
    for g in opt.groups:
        o = off
        for p in g.params:
            k = p.numel()
            if o <= idx < o + k:
                return g.name, names.get(id(p), "?"), idx - o, tuple(p.shape)
            o += (k + 3) // 4 * 4
        off += g.numel
    return "?", "?", idx, ()


def compare(tag, pipe, opt, a, ref):
    bad = 0
    off = 0
    for g in opt.groups:
        x, y = a[off:off + g.numel], ref[off:off + g.numel]
        d = (x - y).abs()
        tol = 1e-5 * float(y.abs().max()) + 1e-12
        if float(d.max()) > tol:
            i = int(d.argmax())
            n_bad = int((d > tol).sum())
            print(f"  MISMATCH {tag} group {g.name}: {n_bad} elements over {tol:.2e}; worst {float(d.max()):.3e} at {where(pipe, opt, off + i)}"
                  f" got {float(x[i]):.6f} ref {float(y[i]):.6f}", flush=True)
            bad += 1
        off += g.numel
    return bad


total = 0
for it in range(n_pipes):
    pipe, opt, rb, batch, rnd = T._setup()
    step = 10_000
    eager = []
    for e in range(3):  # the eager step repeated: pass 0 returns its bias gradients to autograd (joined), later passes sink them
        opt.zero_grad_all()
        outs, ld, _ = pipe.get_train_loss_dict(step, ray_bundle=rb, batch=batch, randoms=rnd)
        total_loss(ld).backward()
        eager.append(opt.flat_g.clone())
        del outs, ld
    torch.cuda.synchronize()
    for e in (1, 2):
        total += compare(f"pipeline {it} eager pass {e} vs pass 0", pipe, opt, eager[e], eager[0])
    stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=2, start_step=step, randoms=rnd)
    stepper.load(rb, batch, sky=rnd["sky_ray_bundle"])
    pipe.model.set_step(step)
    for rep in range(n_rep):
        stepper.graph.replay()
        torch.cuda.synchronize()
        total += compare(f"pipeline {it} replay {rep} vs eager pass 0", pipe, opt, opt.flat_g, eager[0])
    print(f"pipeline {it}: done, mismatches so far {total}", flush=True)
    del stepper, pipe, opt
print("mismatches", total)
