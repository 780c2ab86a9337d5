"""Long run of the graph-replayed train step of the bench workload with a finiteness check of the loss after EVERY step and of every
gradient tensor from step 20 000 on; at the first non-finite value: which model outputs, gradients and parameters are affected.
Round 4 found two such values with it: a multi-view point drawn on the pole of the DDF sphere (step ~2 800) and an
fp16 overflow of a tangent-bearing weight-gradient operand (step ~25 000).     python tools/nan_hunt.py [steps = 60000]   (GPU box)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import torch, bench
from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
from neusky_amd.model_components.losses import total_loss
from util_step import randomise
from test_gpu_step import _module_grads

class Stepper(GraphedTrainStep):
    def _body(self, step):
        self.opt.zero_grad_all()
        outs, loss_dict, metrics = self.pipeline.get_train_loss_dict(step, ray_bundle=self.rb, batch=self.batch, randoms=self.randoms)
        self.outs = outs
        loss = total_loss(loss_dict)
        loss.backward()
        self.opt.collect_grads()
        return loss.detach(), {k: v.detach() for k, v in loss_dict.items()}, metrics

torch.manual_seed(0)
pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
nb = 16
batches = [pipe.datamanager.next_train(i) for i in range(nb)]
skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(nb)]
stepper = Stepper(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
def flat(d, pre=""):
    for k, v in d.items():
        k = str(k)
        if isinstance(v, dict):
            yield from flat(v, pre + k + ".")
        elif torch.is_tensor(v) and v.dtype.is_floating_point:
            yield pre + k, v
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                if torch.is_tensor(x) and x.dtype.is_floating_point:
                    yield f"{pre}{k}[{i}]", x
for i in range(N):
    j = i % nb
    # (the check costs a sync per step)
    stepper.load(batches[j][0], batches[j][1], skies[j])
    pipe.model.set_step(3 + i)
    stepper.graph.replay()
    gbad = [k for k, v in _module_grads(pipe).items() if v is not None and not bool(torch.isfinite(v).all())] if i > 20000 else []
    if gbad or not bool(torch.isfinite(stepper.loss)):
        print("non-finite gradients at step", i, gbad[:14], flush=True)
        eg = stepper.outs.get("eik_grad")
        if torch.is_tensor(eg):
            nrm = eg.detach().norm(dim=-1)
            print("   |eik_grad|: min", float(nrm.min()), "zeros", int((nrm == 0).sum()), "max", float(nrm.max()))
        for k, v in flat(stepper.outs):
            if v.numel() and torch.isfinite(v).all():
                a = v.detach().abs()
                if float(a.max()) > 1e4 or (k in ("weights", "accumulation", "depth", "normal") ):
                    print(f"   finite output {k} {tuple(v.shape)} max {float(a.max()):.3e} min {float(a.min()):.3e}")
        print("first non-finite loss at step", i, {k: float(v) for k, v in stepper.loss_dict.items()}, flush=True)
        for k, v in flat(stepper.outs):
            nf = int((~torch.isfinite(v)).sum())
            if nf:
                idx = torch.nonzero(~torch.isfinite(v.reshape(v.shape[0], -1)).all(dim=1)).flatten()[:8].tolist()
                print(f"   output {k} {tuple(v.shape)}: {nf} non-finite, rows {idx}")
        g = {k: v for k, v in _module_grads(pipe).items() if v is not None}
        print("   non-finite grads:", [k for k, v in g.items() if not torch.isfinite(v).all()][:12])
        bad_p = [n for n, p in pipe.named_parameters() if not torch.isfinite(p).all()]
        print("   non-finite params (before this step's update):", bad_p[:8])
        # the rows of the inputs behind a bad output
        rb = stepper.rb
        for name in ("depth", "expected_termination_dist", "multi_view_expected_termination_dist", "sky_ray_expected_termination_dist", "rgb", "eik_grad"):
            v = stepper.outs.get(name)
            if torch.is_tensor(v) and not torch.isfinite(v).all():
                rows = torch.nonzero(~torch.isfinite(v.reshape(v.shape[0], -1)).all(dim=1)).flatten()[:4]
                print("   ", name, "bad rows", rows.tolist(), "origins", rb.origins.reshape(-1, 3)[rows % rb.origins.reshape(-1, 3).shape[0]].tolist() if rows.numel() else "")
        break
    stepper.opt.all_reduce_gradients()
    stepper.opt.optimizer_scheduler_step_all(3 + i)
    if i % 2000 == 0:
        print(i, float(stepper.loss), flush=True)
else:
    print("no non-finite loss in", N, "steps")
