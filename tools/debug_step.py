import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/tests", ROOT+"/tests/golden"): sys.path.insert(0,p)
import torch
from oracle import neusky_oracle as O
from util_step import *
import test_gpu_step as T
DEV="cuda:0"
torch.manual_seed(0)
R=16
pipe = small_pipeline_config(R=R).setup(device=DEV); pipe.train(); randomise(pipe)
rb, batch = pipe.datamanager.next_train(0)
rnd = make_randoms(pipe, R)
pipe.model.set_step(10_000)
outs, loss_dict, metrics = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(rnd, DEV))
p = oracle_params(pipe)
light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).double()
cfg = oracle_step_cfg(pipe)
ld, out = O.neusky_train_step(p, cfg, rb.origins.cpu().double(), rb.directions.cpu().double(), rb.camera_indices.cpu().reshape(-1), batch["image"].cpu().double(), batch["mask"].cpu(), oracle_randoms(rnd, light), light)
g = outs["visibility_dict"]["sdf_at_termination"].detach().cpu().double().reshape(-1); r = out["sdf_at_termination"].detach().reshape(-1)
print("sdf_at_term maxabs", (g-r).abs().max().item(), "scale", r.abs().max().item())
g = outs["visibility_dict"]["expected_termination_dist"].detach().cpu().double().reshape(-1); r = out["expected_termination_dist"].detach().reshape(-1)
print("t_hat maxabs", (g-r).abs().max().item(), "scale", r.abs().max().item())
xs = (torch.rand(4000,3,generator=torch.Generator().manual_seed(5))*2-1)*1.2
got = pipe.model.field.get_sdf_at_pos(xs.to(DEV)).detach().cpu().double()
ref = O.sdf_at_positions(xs.double(), p, cfg.field_grid).detach()
print("sdf random pts maxabs", (got-ref).abs().max().item(), "scale", ref.abs().max().item())
xd = xs.double().requires_grad_(True)
rr = O.sdf_at_positions(xd, p, cfg.field_grid)
gref = torch.autograd.grad(rr.sum(), xd)[0]
print("oracle |grad sdf| max", gref.norm(dim=-1).max().item(), "mean", gref.norm(dim=-1).mean().item())
xg = xs.to(DEV).requires_grad_(True)
gg = torch.autograd.grad(pipe.model.field.get_sdf_at_pos(xg).sum(), xg)[0].cpu().double()
print("dsdf/dx err", (gg-gref).abs().max().item(), "scale", gref.abs().max().item())
keys = [k for k in p if not k.startswith("reni.")]
which = sys.argv[1:] or list(ld.keys())
for name in which:
    for q in pipe.parameters(): q.grad=None
    outs, loss_dict, _ = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(rnd, DEV))
    if name not in loss_dict: continue
    loss_dict[name].backward()
    try:
        got = T._module_grads(pipe)
    except AttributeError:
        class _Z:
            def view(self,*a): return None
        for q in pipe.parameters():
            if q.grad is None and q.requires_grad: q.grad = torch.zeros_like(q)
        got = T._module_grads(pipe)
    ld, out = O.neusky_train_step(p, cfg, rb.origins.cpu().double(), rb.directions.cpu().double(), rb.camera_indices.cpu().reshape(-1), batch["image"].cpu().double(), batch["mask"].cpu(), oracle_randoms(rnd, light), light)
    grads = torch.autograd.grad(ld[name], [p[k] for k in keys], allow_unused=True)
    bad=[]
    for k, ref in zip(keys, grads):
        gg = got[k]
        if ref is None:
            if gg is not None and gg.abs().max()>0: bad.append((k,'ref None got', gg.abs().max().item()))
            continue
        if gg is None:
            if ref.abs().max()>0: bad.append((k,'got None ref', ref.abs().max().item()))
            continue
        a,b = gg.detach().cpu().double().reshape(-1), ref.reshape(-1)
        sc=b.abs().max().item(); err=(a-b).abs().max().item()
        if err > 1e-3*sc+1e-14: bad.append((k, f"{err:.3e}", f"{sc:.3e}"))
    print("LOSS", name, float(loss_dict[name]), float(ld[name]), "BAD:", bad)
