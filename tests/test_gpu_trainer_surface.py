"""The drop-in boundary, driven the way nerfstudio's Trainer drives it (VERDICT r5 "missing" 1; neusky/pipelines/neusky_pipeline.py
:198-200 = the DDP wrap inside the pipeline, :241-291 = get_train_loss_dict, the only thing the trainer calls): zero_grad ->
get_train_loss_dict -> sum -> backward -> torch.optim.Adam(eps=1e-15).step per group -> scheduler.step, with nothing from
neusky_amd.engine.  The gradient exchange and the HIP-graph replay are the pipeline's own.

 (a) world_size 2 (two processes on cuda:0, gloo): exchanged gradients and parameters after 4 steps (2 eager + 2 replayed) equal the
     one-rank run on the concatenated 32-ray batch;
 (b) world_size 1 over RCCL ("nccl"): the same loop with the exchange forced on over a one-rank communicator equals the run without;
 (c) the trainer loop with graph replay + torch Adam lands where engine.GraphedTrainStep (same replay + fused nsky_adam_step) lands
     after 8 steps."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
for p_ in (HERE, os.path.join(HERE, "golden")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(world, backend, tmp, steps=4, eager_steps=2, tag=""):
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [os.path.join(tmp, f"t{tag}{world}_{backend}_r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "trainer_loop_worker.py"), str(r), str(world), str(port), outs[r], backend,
                               str(steps), str(eager_steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(f"--- rank {i} (exit {p.returncode})\n{l[-3000:]}" for i, (p, l) in enumerate(zip(procs, logs)))
    return np.load(outs[0])


def _lr_of(name):
    """the learning rate of the group a parameter belongs to (neusky_config.py:216-237), for the parameter bars"""
    if ".proposal_networks." in name:
        return 1e-2
    if "illumination_latents" in name or name.endswith("train_scale"):
        return 1e-2
    if ".visibility_field." in name:
        return 1e-4
    return 1e-3


def _rel_bar(name):
    """relative L2 bar on a tensor's movement.  The hash tables' gradients are atomic scatters (summation order differs from launch to
    launch) and most of their touched rows see a handful of nearly cancelling contributions: under Adam(eps = 1e-15) a few per cent of
    those rows step the other way between ANY two runs (measured: 3 % / 7 % between two replays of the same graph with different
    optimizers); every other tensor is a long, well-conditioned sum."""
    return 0.15 if name.endswith("encoding.params") else 2e-2


def _compare_gradients(a, b, prefix, rel=2e-3):
    keys = sorted(k for k in b.files if k.startswith(prefix))
    assert keys == sorted(k for k in a.files if k.startswith(prefix)) and len(keys) > 40
    bad = []
    for k in keys:
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        scale = np.abs(y).max()
        if np.abs(x - y).max() > rel * scale + 1e-9:  # fp32 reduction-order noise only (tests/test_gpu_two_ranks.py's bar)
            bad.append((k, float(np.abs(x - y).max()), float(scale)))
    assert not bad, bad


def _compare_parameters(a, b, steps, what):
    """Adam with eps = 1e-15 turns a gradient into a step of ~lr whatever its size, so an element whose gradient is reduction-order
    noise around zero (a hash-table row a few cancelling samples touch) may step the other way: the bar is on the MOVEMENT -- every
    tensor's movement agrees in the bulk (relative L2 <= 2 %, at most 0.1 % of the elements off by more than 5 % of the lr * steps a
    parameter can travel) -- not on the worst element."""
    keys = sorted(k for k in b.files if k.startswith("p:"))
    assert keys == sorted(k for k in a.files if k.startswith("p:"))
    bad, moved = [], 0
    for k in keys:
        d_a = a[k].astype(np.float64) - a["b:" + k[2:]].astype(np.float64)
        d_b = b[k].astype(np.float64) - b["b:" + k[2:]].astype(np.float64)
        nb = np.linalg.norm(d_b)
        if nb == 0.0:
            assert np.linalg.norm(d_a) == 0.0, k
            continue
        moved += 1
        travel = _lr_of(k) * steps
        off = float((np.abs(d_a - d_b) > 0.05 * travel).mean())
        rel = float(np.linalg.norm(d_a - d_b) / nb)
        if rel > _rel_bar(k) or off > 1e-3:
            bad.append((k, rel, off))
    assert moved > 40 and not bad, (what, moved, bad)


def test_trainer_loop_two_ranks_equal_one_rank(tmp_path):
    two = _launch(2, "gloo", str(tmp_path))
    one = _launch(1, "none", str(tmp_path))
    assert bool(two["in_slab"]), "p.grad are not views of the pipeline's slab after the exchange"
    assert int(two["moved"]) > 40 and int(one["moved"]) > 40
    _compare_gradients(two, one, "g0:")  # the first, eager pass: exchanged by the end-of-pass hook
    _compare_parameters(two, one, 4, "2 ranks vs 1")
    # batch-mean losses of half batches are not the concatenated batch's; they are finite and move
    assert np.isfinite(two["losses"]).all() and np.isfinite(one["losses"]).all()


def test_trainer_loop_one_rank_over_rccl(tmp_path):
    rccl = _launch(1, "nccl", str(tmp_path))
    plain = _launch(1, "none", str(tmp_path), tag="p")
    assert bool(rccl["in_slab"])
    _compare_gradients(rccl, plain, "g0:", rel=1e-4)  # an AVG all-reduce over one rank is the identity: only the hash-table atomics' order differs
    _compare_parameters(rccl, plain, 4, "1 rank over RCCL vs no process group")
    assert np.allclose(rccl["losses"], plain["losses"], rtol=1e-4)


def test_trainer_loop_with_graph_replay_lands_where_the_engine_lands():
    """8 steps on injected draws: (nerfstudio loop, graph_replay, torch Adam) vs engine.GraphedTrainStep (same TrainGraph, fused Adam)"""
    import torch
    from trainer_loop_worker import STEP0, nerfstudio_train_iteration, torch_optimizers
    from two_rank_worker import build, shard
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    dev = "cuda:0"
    runs = {}
    for mode in ("trainer", "trainer_slab_adam", "engine"):
        pipe, rb, batch, rnd = build(dev, 1, 0)
        rbs, bs, rs = shard(rb, batch, rnd, 1, 0, dev)
        before = {n: p.detach().clone() for n, p in pipe.named_parameters() if p.requires_grad}
        losses = []
        if mode.startswith("trainer"):
            pipe.config.graph_replay, pipe.config.graph_replay_warmup = True, 0
            opts, scheds = torch_optimizers(pipe, fused=mode.endswith("slab_adam"))  # (SlabAdam: what plugin.SlabAdamOptimizerConfig gives ns-train)
            for i in range(8):
                losses.append(float(nerfstudio_train_iteration(pipe, opts, scheds, STEP0 + i, ray_bundle=rbs, batch=bs, randoms=rs)))
            assert pipe._train_graph is not None
        else:
            opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
            stepper = GraphedTrainStep(pipe, opt, rbs, bs, warmup=1, start_step=STEP0, randoms=rs)
            for i in range(8):
                losses.append(float(stepper.step(STEP0 + i, rbs, bs, rs["sky_ray_bundle"])[0]))
        torch.cuda.synchronize()
        runs[mode] = (losses, {n: (p.detach() - before[n]).double().cpu() for n, p in pipe.named_parameters() if p.requires_grad})
        del pipe
    for which in ("trainer", "trainer_slab_adam"):
        _lands_where(runs[which], runs["engine"], which)


def _lands_where(trainer_run, engine_run, which):
    lt, le = trainer_run[0], engine_run[0]
    assert np.allclose(lt, le, rtol=2e-4), (which, lt, le)
    bad, moved = [], 0
    for n, d_e in engine_run[1].items():
        d_t = trainer_run[1][n]
        if float(d_e.norm()) == 0.0:
            assert float(d_t.norm()) == 0.0, n
            continue
        moved += 1
        travel = _lr_of(n) * 8
        rel = float((d_t - d_e).norm() / d_e.norm())
        off = float(((d_t - d_e).abs() > 0.05 * travel).double().mean())
        if rel > _rel_bar(n) or off > 1e-3:
            bad.append((n, rel, off))
    assert moved > 40 and not bad, (which, bad)
