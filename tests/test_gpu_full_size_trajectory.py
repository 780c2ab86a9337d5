"""Train-trajectory parity AT THE BENCH'S SIZE (neusky/pipelines/neusky_pipeline.py:241-291,
neusky/configs/neusky_config.py:216-237): three consecutive graph-replayed training steps of the 1024 rays x 96 samples x 512
directions workload -- the launches, grids and workgroup rounds the bench line times -- against the oracle's training steps with a
float64 Adam.

The oracle cannot evaluate 1024 rays x 256 DDF queries in seconds, and the step's objective couples every ray (batch means, the
interlevel and eikonal sums), so the slice trick of test_gpu_full_size.py (a probe on 16 rays) does not carry over to a training step.
What does: REPLICATION.  The batch is 16 distinct rays repeated 64 times -- origins, directions, cameras, pixels, masks and every
per-ray random draw alike -- the 1024 DDF-fit rays are 16 distinct vMF rays x 64, the 256 sky rays 16 x 16.  Every term of the
objective is a mean (or a sum over a mean's support divided by its count), so the replicated batch has the objective AND the
gradient of the 16-ray batch exactly, term by term, while the HIP path launches the full-size kernels on 1024 rays / 263 456 DDF rows
/ 99 304 field points (a hash-table row touched by a ray is touched 64 times: the scatter's long sums are exercised, not avoided).
The oracle runs the 16-ray batch.

Bars as in tests/test_gpu_trajectory.py (Adam with eps = 1e-15 turns fp32 noise around zero into steps of ~lr, so the yardstick is
the oracle itself in float32 on the same inputs): loss trace step 0 within 1e-5 relative and every TERM within 2e-4; step i within
max(2e-4, 4 x the float32 oracle's distance); parameters after the last step: rms distance from the float64 trajectory in units of
the tensor's possible travel <= max(0.01, 3 x the float32 oracle's)."""
import os
import sys

import pytest
import torch

from oracle import neusky_oracle as O
from util_step import make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randomise, randoms_to

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K, STEP0, NS = 3, 10_000, 16


def _rep(t, n):
    return t.repeat(n, *([1] * (t.dim() - 1))).contiguous()


@pytest.fixture(scope="module")
def run():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from test_gpu_trajectory import _group_of
    from neusky_amd.cameras.rays import RayBundle
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    torch.manual_seed(0)
    pipe = bench.build_pipeline(DEV, 1, 0)  # BASELINE configs[2] at full size
    randomise(pipe)
    R = bench.RAYS
    rep = R // NS
    sc = pipe.config.visibility_train_sampler
    n_fit = sc.num_samples_on_sphere * sc.num_rays_per_sample
    n_sky = pipe.config.num_sky_rays
    assert R % NS == 0 and n_fit % NS == 0 and n_sky % NS == 0
    opt_cfg = neusky_optimizers()
    opt = Optimizers(opt_cfg, pipe.get_param_groups())
    cfg = oracle_step_cfg(pipe)

    small, full = [], []  # per step: the 16-ray problem of the oracle, its 64-fold replication for the HIP path
    for i in range(K):
        rb, batch = pipe.datamanager.next_train(i)
        r = make_randoms(pipe, R, seed=300 + i)
        s_rb = dict(o=rb.origins[:NS].cpu(), d=rb.directions[:NS].cpu(), cam=rb.camera_indices[:NS].cpu(), img=batch["image"][:NS].cpu(),
                    mask=batch["mask"][:NS].cpu())
        s_rnd = dict(r)
        s_rnd["jitters"] = [j[:NS] for j in r["jitters"]]
        s_rnd["ddf_jitters"] = [j[:NS] for j in r["ddf_jitters"]]
        s_rnd["ddf_rays"] = tuple(t[:NS] for t in r["ddf_rays"])
        s_rnd["mv_points"] = r["mv_points"][:NS]
        sky = r["sky_ray_bundle"]
        s_rnd["sky_ray_bundle"] = RayBundle(origins=sky.origins[:NS].contiguous(), directions=sky.directions[:NS].contiguous())
        small.append((s_rb, s_rnd))
        f_rb = RayBundle(origins=_rep(s_rb["o"], rep).to(DEV), directions=_rep(s_rb["d"], rep).to(DEV), pixel_area=_rep(rb.pixel_area[:NS], rep),
                         camera_indices=_rep(s_rb["cam"], rep).to(DEV), metadata={k: _rep(v[:NS], rep) for k, v in rb.metadata.items()})
        f_batch = {"image": _rep(s_rb["img"], rep).to(DEV), "mask": _rep(s_rb["mask"], rep).to(DEV)}
        f_rnd = dict(r)
        f_rnd["jitters"] = [_rep(j, rep) for j in s_rnd["jitters"]]
        f_rnd["ddf_jitters"] = [_rep(j, n_fit // NS) for j in s_rnd["ddf_jitters"]]
        f_rnd["ddf_rays"] = tuple(_rep(t, n_fit // NS) for t in s_rnd["ddf_rays"])
        f_rnd["mv_points"] = _rep(s_rnd["mv_points"], n_fit // NS)
        f_rnd["sky_ray_bundle"] = RayBundle(origins=_rep(sky.origins[:NS], n_sky // NS).to(DEV), directions=_rep(sky.directions[:NS], n_sky // NS).to(DEV))
        full.append((f_rb, f_batch, f_rnd))
    p0 = {k: v.detach().clone() for k, v in oracle_params(pipe).items()}

    def dev_rnd(r):
        d = randoms_to(r, DEV)
        for k in ("light_rotation", "grid_perturb", "grid_dirs"):
            d[k] = d[k].to(DEV)
        return d

    # ---- HIP: ONE captured graph of the full-size step, K replays, nsky_adam_step
    stepper = GraphedTrainStep(pipe, opt, full[0][0], full[0][1], warmup=2, start_step=STEP0, randoms=dev_rnd(full[0][2]))
    hip_trace, hip_terms = [], []
    for i in range(K):
        f_rb, f_batch, f_rnd = full[i]
        stepper.tg.load_randoms({k: v for k, v in dev_rnd(f_rnd).items() if k != "sky_ray_bundle"})
        loss, ld, _ = stepper.step(STEP0 + i, f_rb, f_batch, f_rnd["sky_ray_bundle"])
        hip_trace.append(float(loss))
        hip_terms.append({k: float(v) for k, v in ld.items()})
    torch.cuda.synchronize()

    # ---- oracle on the 16-ray batch: autograd + Adam, float64 and float32
    def oracle_run(dt):
        q = {k: v.detach().to(dt).clone().requires_grad_(True) for k, v in p0.items()}
        keys = [k for k in q if not k.startswith("reni.")]
        m_ = {k: torch.zeros_like(q[k]) for k in keys}
        v_ = {k: torch.zeros_like(q[k]) for k in keys}
        trace, terms, travel = [], [], {k: 0.0 for k in keys}
        for i in range(K):
            s_rb, s_rnd = small[i]
            light = pipe.model.illumination_sampler(rotation=s_rnd["light_rotation"]).to(dt)
            ld, _ = O.neusky_train_step(q, cfg, s_rb["o"].to(dt), s_rb["d"].to(dt), s_rb["cam"].reshape(-1), s_rb["img"].to(dt), s_rb["mask"],
                                        oracle_randoms(s_rnd, light, dt), light)
            loss = sum(ld.values())
            grads = torch.autograd.grad(loss, [q[k] for k in keys], allow_unused=True)
            trace.append(float(loss.detach()))
            terms.append({k: float(v.detach()) for k, v in ld.items()})
            with torch.no_grad():
                for k, g in zip(keys, grads):
                    oc = opt_cfg[_group_of(k)]
                    a, sched = oc["optimizer"], oc["scheduler"]
                    lr = a.lr * sched.factor(STEP0 + i)
                    if g is None:
                        g = torch.zeros_like(q[k])
                    travel[k] += lr
                    m_[k].mul_(a.betas[0]).add_(g, alpha=1 - a.betas[0])
                    v_[k].mul_(a.betas[1]).addcmul_(g, g, value=1 - a.betas[1])
                    mhat, vhat = m_[k] / (1 - a.betas[0] ** (i + 1)), v_[k] / (1 - a.betas[1] ** (i + 1))
                    q[k].sub_(lr * mhat / (vhat.sqrt() + a.eps))
        return q, keys, trace, terms, travel

    p, keys, ref_trace, ref_terms, travel = oracle_run(torch.float64)
    p32, _, f32_trace, _, _ = oracle_run(torch.float32)
    return dict(pipe=pipe, p=p, p32=p32, p0=p0, keys=keys, hip_trace=hip_trace, ref_trace=ref_trace, f32_trace=f32_trace,
                hip_terms=hip_terms, ref_terms=ref_terms, travel=travel)


def test_full_size_loss_trace_follows_the_oracle(run):
    os.makedirs("gpurun_out", exist_ok=True)
    rel = lambda a, b: abs(a - b) / abs(b)  # noqa: E731
    env, bars = 0.0, []
    for a32, b in zip(run["f32_trace"], run["ref_trace"]):
        env = max(env, rel(a32, b))
        bars.append(max(2e-4, 4.0 * env))
    bars[0] = 1e-5
    with open("gpurun_out/r06_full_size_trajectory.txt", "w") as f:
        f.write(f"{K} consecutive graph-replayed training steps from step {STEP0} at 1024 rays x 96 samples x 512 directions "
                f"(16 distinct rays x 64; oracle on the 16)\nstep: objective HIP | float64 oracle | float32 oracle | rel distance from float64: HIP, float32 oracle | bar\n")
        for i, (a, b, c) in enumerate(zip(run["hip_trace"], run["ref_trace"], run["f32_trace"])):
            f.write(f"step {i}: {a:.8f} {b:.8f} {c:.8f}  {rel(a, b):.2e} {rel(c, b):.2e}  {bars[i]:.1e}\n")
        f.write("step 0, term by term: HIP | float64 oracle\n")
        for k in run["ref_terms"][0]:
            f.write(f"  {k:28s} {run['hip_terms'][0][k]:.8e} {run['ref_terms'][0][k]:.8e}\n")
    for i, (a, b) in enumerate(zip(run["hip_trace"], run["ref_trace"])):
        assert rel(a, b) <= bars[i], (i, run["hip_trace"], run["ref_trace"], run["f32_trace"])
    for k in run["ref_terms"][0]:
        a, b = run["hip_terms"][0][k], run["ref_terms"][0][k]
        assert abs(a - b) <= 2e-4 * max(abs(b), 1e-3), (k, a, b)
    assert all(sorted(ta) == sorted(tb) for ta, tb in zip(run["hip_terms"], run["ref_terms"]))


def test_full_size_parameters_after_the_steps(run):
    from test_gpu_trajectory import _module_params
    got = _module_params(run["pipe"])
    rows, bad = [], []
    rms = lambda t: float(t.double().pow(2).mean().sqrt())  # noqa: E731
    for k in run["keys"]:
        T = run["travel"][k]
        ref = run["p"][k].detach()
        d_hip, d_f32 = got[k] - ref, run["p32"][k].detach().double() - ref
        moved = ref - run["p0"][k]
        touched = moved != 0
        if int(touched.sum()) == 0:
            continue
        r_hip, r_f32 = rms(d_hip[touched]) / T, rms(d_f32[touched]) / T
        worst = float(d_hip.abs().max()) / T
        bar = max(0.01, 3.0 * r_f32)
        rows.append((k, int(touched.sum()), rms(moved[touched]) / T, r_hip, r_f32, worst, bar))
        if r_hip > bar or worst > 2.0 + 1e-6:
            bad.append((k, r_hip, r_f32, worst))
    with open("gpurun_out/r06_full_size_trajectory.txt", "a") as f:
        f.write("\nparameters after the last step, in units of the tensor's possible travel T = sum of the steps' learning rates:\n"
                "tensor  touched elements  rms displacement | rms distance from the float64 trajectory: HIP, float32 oracle | largest HIP | bar (rms)\n")
        for r in rows:
            f.write(f"{r[0]:24s} {r[1]:9d}  {r[2]:.3f} | {r[3]:.3e} {r[4]:.3e} | {r[5]:.3e} | {r[6]:.2e}\n")
    assert not bad, bad
    assert len(rows) > 50
