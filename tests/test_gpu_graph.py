"""The HIP-graph-captured training iteration: replays run, are finite, move every parameter group, and agree with the
eager path on a fixed input when the in-graph random draws are irrelevant (statistically: same loss scale)."""
import pytest
import torch

from util_step import randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_graphed_train_step_replays():
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers, train_iteration
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, vmf=(2, 16), sky=16, images=7).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    batches = [pipe.datamanager.next_train(i) for i in range(8)]
    eager_loss, _, _ = train_iteration(pipe, opt, 100, ray_bundle=batches[0][0], batch=batches[0][1])
    before = {g.name: g.flat_p.clone() for g in opt.groups}
    stepper = GraphedTrainStep(pipe, opt, batches[1][0], batches[1][1], warmup=2, start_step=101)
    losses = []
    for i in range(5):
        rb, b = batches[2 + i]
        loss, loss_dict, _ = stepper.step(110 + i, rb, b)
        losses.append(float(loss))
        assert all(torch.isfinite(v) for v in loss_dict.values())
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert 0.2 < losses[0] / float(eager_loss) < 5.0, (losses, float(eager_loss))
    for g in opt.groups:  # every group received gradients through the graph and was stepped
        assert (g.flat_p - before[g.name]).abs().max().item() > 0, g.name
    # different inputs give different losses: the static input buffers really feed the graph
    assert len({round(l, 6) for l in losses}) > 1


def _setup(R=64):
    from neusky_amd.engine import Optimizers, neusky_optimizers
    from util_step import make_randoms, randoms_to
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=R, num_prop=(32, 16), S=16, D=128, vmf=(2, 16), sky=16, images=7).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    rb, batch = pipe.datamanager.next_train(0)
    rnd = randoms_to(make_randoms(pipe, R), DEV)
    rnd["light_rotation"] = rnd["light_rotation"].to(DEV)
    rnd["grid_perturb"], rnd["grid_dirs"] = rnd["grid_perturb"].to(DEV), rnd["grid_dirs"].to(DEV)
    return pipe, opt, rb, batch, rnd


def test_graph_replay_equals_eager_on_injected_randoms():
    """with every random draw of the step injected (static device buffers the graph reads), a replayed
    step must reproduce the eager step: loss terms to 1e-6 relative, every gradient to 1e-5 of its group's max (float atomics
    make the hash-table / split-K sums order dependent, so not bit for bit).  The eager step itself is checked against the
    float64 oracle at this size in test_gpu_step.py (step_big), so this pins graph replay to the oracle transitively."""
    from neusky_amd.engine import GraphedTrainStep
    from neusky_amd.model_components.losses import total_loss
    pipe, opt, rb, batch, rnd = _setup()
    step = 10_000
    opt.zero_grad_all()
    outs, ld, _ = pipe.get_train_loss_dict(step, ray_bundle=rb, batch=batch, randoms=rnd)
    total_loss(ld).backward()
    eager_ld = {k: float(v.detach()) for k, v in ld.items()}
    eager_rgb = outs["rgb"].detach().clone()
    eager_inds = [t.clone() for t in outs["pdf_inds_list"]]
    eager_g = opt.flat_g.clone()
    del outs, ld  # the eager autograd graph (and its AccumulateGrad nodes, bound to the default stream) must be gone before the capture
    stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=2, start_step=step, randoms=rnd)
    stepper.load(rb, batch, sky=rnd["sky_ray_bundle"])  # the graph owns its sky-ray buffers: same sky rays as the eager step
    pipe.model.set_step(step)
    for rep in range(2):
        stepper.graph.replay()
        torch.cuda.synchronize()
        for k, v in stepper.loss_dict.items():
            assert abs(float(v) - eager_ld[k]) <= 1e-6 * max(abs(eager_ld[k]), 1e-3), (rep, k, float(v), eager_ld[k])
        off = 0
        for g in opt.groups:
            a, b = opt.flat_g[off:off + g.numel], eager_g[off:off + g.numel]
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (rep, g.name)
            off += g.numel


def test_first_pass_gradients_wait_for_the_side_stream():
    """the FIRST backward pass of a pipeline hands its bias accumulators to autograd (they are slab-resident only from the second pass on)
    while the weight-gradient kernels that fill them run on the side stream (ops.async_weight_gradients): every node sharing such an
    accumulator joins before it returns (ops.join_unless_sunk).  Round 5: one first pass in eight lost a node's share of `glin0.bias`.
    Pass 0 against pass 1 (sunk biases, joined at the end of the pass) on the same injected inputs, on several fresh pipelines."""
    from neusky_amd.model_components.losses import total_loss
    for it in range(5):
        pipe, opt, rb, batch, rnd = _setup()
        passes = []
        for _ in range(2):
            opt.zero_grad_all()
            outs, ld, _ = pipe.get_train_loss_dict(10_000, ray_bundle=rb, batch=batch, randoms=rnd)
            total_loss(ld).backward()
            passes.append(opt.flat_g.clone())
            del outs, ld
        torch.cuda.synchronize()
        off = 0
        for g in opt.groups:
            a, b = passes[0][off:off + g.numel], passes[1][off:off + g.numel]
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, (it, g.name, int((a - b).abs().argmax()))
            off += g.numel
        del pipe, opt


def test_proposal_anneal_follows_the_step_under_graph_replay():
    """the proposal-weight anneal is a device scalar the graph READS; nothing inside the capture may write
    it.  Early steps must re-sample near-uniformly (anneal ~ 0.05), late steps with anneal 1, exactly as the eager step."""
    from neusky_amd.engine import GraphedTrainStep
    pipe, opt, rb, batch, rnd = _setup()
    sampler = pipe.model.proposal_sampler
    stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=2, start_step=500, randoms=rnd)
    stepper.load(rb, batch, sky=rnd["sky_ray_bundle"])
    seen = {}
    for s in (5, 2000):
        pipe.model.set_step(s)
        stepper.graph.replay()
        torch.cuda.synchronize()
        seen[s] = (float(sampler._anneal_t), float(stepper.loss))
    bias = lambda x, b: (b * x) / ((b - 1) * x + 1)  # noqa: E731
    assert abs(seen[5][0] - bias(5 / 1000, 10.0)) < 1e-6 and seen[2000][0] == 1.0, seen
    # and the replayed losses equal the eager ones at the same steps (same injected randoms)
    from neusky_amd.model_components.losses import total_loss
    for s in (5, 2000):
        _, ld, _ = pipe.get_train_loss_dict(s, ray_bundle=rb, batch=batch, randoms=rnd)
        e = float(total_loss(ld))
        assert abs(e - seen[s][1]) <= 1e-5 * abs(e), (s, e, seen[s][1])
    assert abs(seen[5][1] - seen[2000][1]) > 1e-6 * abs(seen[5][1])  # the anneal really changes the step


def test_icosphere_direction_set_trains_and_replays():
    """the reference's own direction set (neusky_config.py:97-101 -> illumination_samplers.py:85-119: an icosphere; order 8 = 642
    vertices here, the nearest to its num_directions = 512) instead of the default antipodal lattice: the set is centrally symmetric,
    so exactly D / 2 = 321 directions lie in the upper hemisphere after any rotation -- static shapes, the graph path applies -- and the
    step's radiance matches the float64 oracle on the same (rotated) set; a replayed graph reproduces the eager step"""
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    from neusky_amd.model_components.losses import total_loss
    from oracle import neusky_oracle as O
    from util_step import make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randoms_to
    torch.manual_seed(0)
    cfg = small_pipeline_config(R=32, num_prop=(32, 16), S=12, D=32, vmf=(2, 16), sky=16, images=5)
    cfg.model.illumination_sampler.icosphere_order = 8
    pipe = cfg.setup(device=DEV)
    pipe.train()
    randomise(pipe)
    D = pipe.model.illumination_sampler.directions.shape[0]
    assert D == 642
    base = pipe.model.illumination_sampler.directions
    assert (torch.cdist(base.double(), -base.double()).min(dim=1).values < 1e-5).all(), "the icosphere is centrally symmetric"
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    rb, batch = pipe.datamanager.next_train(0)
    R = 32
    rnd_host = make_randoms(pipe, R)
    rnd = randoms_to(rnd_host, DEV)
    for k in ("light_rotation", "grid_perturb", "grid_dirs"):
        rnd[k] = rnd[k].to(DEV)
    step = 10_000
    pipe.model.set_step(step)
    opt.zero_grad_all()
    outs, ld, _ = pipe.get_train_loss_dict(step, ray_bundle=rb, batch=batch, randoms=rnd)
    assert outs["visibility_dict"]["visibility"].shape == (R, D)
    assert outs["visibility_dict"]["expected_termination_dist"].numel() == R * (D // 2)
    total_loss(ld).backward()
    eager_ld = {k: float(v.detach()) for k, v in ld.items()}
    eager_g = opt.flat_g.clone()
    got = outs["rgb"].detach().cpu().double()
    del outs, ld
    p = oracle_params(pipe)
    light = pipe.model.illumination_sampler(rotation=rnd_host["light_rotation"]).double()
    ref_ld, ref = O.neusky_train_step(p, oracle_step_cfg(pipe), rb.origins.cpu().double(), rb.directions.cpu().double(),
                                      rb.camera_indices.cpu().reshape(-1), batch["image"].cpu().double(), batch["mask"].cpu(),
                                      oracle_randoms(rnd_host, light), light)
    rel = ((got - ref["rgb"].detach()).abs().max() / ref["rgb"].detach().abs().max()).item()
    assert rel < 1e-4, rel
    for k, v in ref_ld.items():
        assert abs(eager_ld[k] - float(v)) < 2e-4 * max(abs(float(v)), 1e-3), (k, eager_ld[k], float(v))
    stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=2, start_step=step, randoms=rnd)
    stepper.load(rb, batch, sky=rnd["sky_ray_bundle"])
    pipe.model.set_step(step)
    stepper.graph.replay()
    torch.cuda.synchronize()
    for k, v in stepper.loss_dict.items():
        assert abs(float(v) - eager_ld[k]) <= 1e-6 * max(abs(eager_ld[k]), 1e-3), (k, float(v), eager_ld[k])
    assert float((opt.flat_g - eager_g).abs().max()) <= 1e-5 * float(eager_g.abs().max())
