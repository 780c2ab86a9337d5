"""radiance / distance parity of the HIP forward against the float64 oracle for the active NSKY_PRECISION (GPU box)"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/tests", ROOT+"/tests/golden"): sys.path.insert(0,p)
import torch
from oracle import neusky_oracle as O
from util_step import *
DEV="cuda:0"
for seed in (0, 1):
    torch.manual_seed(seed)
    R=32
    pipe = small_pipeline_config(R=R, D=48).setup(device=DEV); pipe.train(); randomise(pipe, seed=seed)
    rb, batch = pipe.datamanager.next_train(0)
    rnd = make_randoms(pipe, R, seed=seed)
    pipe.model.set_step(10_000)
    with torch.no_grad():
        pass
    outs, loss_dict, _ = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(rnd, DEV))
    p = oracle_params(pipe)
    light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).double()
    ld, out = O.neusky_train_step(p, oracle_step_cfg(pipe), rb.origins.cpu().double(), rb.directions.cpu().double(), rb.camera_indices.cpu().reshape(-1), batch["image"].cpu().double(), batch["mask"].cpu(), oracle_randoms(rnd, light), light)
    rel = ((outs["rgb"].detach().cpu().double()-out["rgb"]).abs().max()/out["rgb"].abs().max()).item()
    th = (outs["visibility_dict"]["expected_termination_dist"].detach().cpu().double()-out["expected_termination_dist"]).abs().max().item()
    inds = all(torch.equal(a.cpu().long(), b) for a,b in zip(outs["pdf_inds_list"], out["pdf_inds_list"]))
    lmax = max(abs(float(loss_dict[k])-float(ld[k]))/max(abs(float(ld[k])),1e-3) for k in ld)
    print(f"seed {seed}: radiance rel {rel:.2e}  t_hat abs {th:.2e}  inds_exact {inds}  worst loss rel {lmax:.2e}")
