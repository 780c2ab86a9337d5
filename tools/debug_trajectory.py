"""one Adam step, HIP vs float64 / float32 oracle: which elements move differently, and what their gradients look like (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import torch
from oracle import neusky_oracle as O
from util_step import make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randomise, randoms_to, small_pipeline_config
from test_gpu_step import _module_grads
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration

DEV = "cuda:0"
R = 64
torch.manual_seed(0)
pipe = small_pipeline_config(R=R, num_prop=(32, 16), S=16, D=128, vmf=(2, 16), sky=16, images=7).setup(device=DEV)
pipe.train(); randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
rb, batch = pipe.datamanager.next_train(0)
rnd = make_randoms(pipe, R, seed=100)
p0 = {k: v.detach().clone() for k, v in oracle_params(pipe).items()}
cfg = oracle_step_cfg(pipe)
d = randoms_to(rnd, DEV)
for k in ("light_rotation", "grid_perturb", "grid_dirs"):
    d[k] = d[k].to(DEV)
pipe.model.set_step(10_000)
train_iteration(pipe, opt, 10_000, ray_bundle=rb, batch=batch, randoms=d)
torch.cuda.synchronize()
g_hip = {k: (v.detach().cpu().double().clone() if v is not None else None) for k, v in _module_grads(pipe).items()}
p1 = {k: v.detach().cpu().double() for k, v in oracle_params(pipe).items()}


def oracle_grads(dt):
    q = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in p0.items()}
    light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).to(dt)
    ld, _ = O.neusky_train_step(q, cfg, rb.origins.cpu().to(dt), rb.directions.cpu().to(dt), rb.camera_indices.cpu().reshape(-1),
                                batch["image"].cpu().to(dt), batch["mask"].cpu(), oracle_randoms(rnd, light, dt), light)
    keys = [k for k in q if not k.startswith("reni.")]
    gs = torch.autograd.grad(sum(ld.values()), [q[k] for k in keys], allow_unused=True)
    return dict(zip(keys, gs))


g64, g32 = oracle_grads(torch.float64), oracle_grads(torch.float32)
for k in ("prop0.table", "prop1.table", "field.table", "field.clin0.v", "ddf.film_w1", "ddf.map_wo", "field.glin0.v"):
    a, b, c = g_hip[k].reshape(-1), g64[k].reshape(-1), g32[k].double().reshape(-1)
    dp = (p1[k] - p0[k]).reshape(-1)
    mx = b.abs().max()
    sign_hip = (a.sign() != b.sign()) & ((a != 0) | (b != 0))
    sign_f32 = (c.sign() != b.sign()) & ((c != 0) | (b != 0))
    nz = int((b != 0).sum())
    print(f"{k}: {b.numel()} elements, {nz} with nonzero f64 gradient, max |g| {mx:.3e}")
    print(f"   sign(g) differs from f64: HIP {int(sign_hip.sum())}, f32 oracle {int(sign_f32.sum())};  HIP nonzero where f64 zero {int(((a != 0) & (b == 0)).sum())}, "
          f"HIP zero where f64 nonzero {int(((a == 0) & (b != 0)).sum())}; f32: {int(((c != 0) & (b == 0)).sum())} / {int(((c == 0) & (b != 0)).sum())}")
    for lo, hi in ((0, 1e-12), (1e-12, 1e-9), (1e-9, 1e-7), (1e-7, 1e-5), (1e-5, 1e-3), (1e-3, 2.0)):
        sel = (b.abs() / mx >= lo) & (b.abs() / mx < hi) & (b != 0)
        n = int(sel.sum())
        if n:
            print(f"   |g|/max in [{lo:.0e}, {hi:.0e}): {n:8d} elements, sign flips HIP {int((sign_hip & sel).sum()):7d} f32 {int((sign_f32 & sel).sum()):7d}; "
                  f"median |g_hip - g64| / max {float(((a - b).abs()[sel] / mx).median()):.2e}, f32 {float(((c - b).abs()[sel] / mx).median()):.2e}; moved {int((dp[sel] != 0).sum())}")

print("---- entries with a HIP gradient where the float64 gradient is exactly zero")
for k in ("prop0.table", "prop1.table"):
    a, b = g_hip[k].reshape(-1), g64[k].reshape(-1)
    sel = (a != 0) & (b == 0)
    v = a[sel].abs()
    if v.numel() == 0:
        print(k, "none")
        continue
    print(k, int(sel.sum()), "|g_hip| min / median / max", float(v.min()), float(v.median()), float(v.max()), "max |g64|", float(b.abs().max()))
    rows = torch.nonzero(sel.reshape(-1, 2).any(1))[:, 0]
    print("   table rows:", rows[:12].tolist(), "...", "moved by", (p1[k] - p0[k]).reshape(-1)[sel].abs().median().item())
# the interlevel term's own inputs: per-level weights of both evaluations
print("---- all tensors: HIP nonzero where f64 zero | HIP zero where f64 nonzero | same for the f32 oracle")
for k in g64:
    if g64[k] is None or g_hip.get(k) is None:
        continue
    a, b, c = g_hip[k].reshape(-1), g64[k].reshape(-1), g32[k].double().reshape(-1)
    print(f"{k:24s} {int(((a != 0) & (b == 0)).sum()):8d} {int(((a == 0) & (b != 0)).sum()):8d} | {int(((c != 0) & (b == 0)).sum()):8d} {int(((c == 0) & (b != 0)).sum()):8d}")
