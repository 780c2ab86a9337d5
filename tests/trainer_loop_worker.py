"""One rank of tests/test_gpu_trainer_surface.py: nerfstudio's training loop, verbatim in its order --

    for optimizer: optimizer.zero_grad()            # Optimizers.zero_grad_all (torch: .grad = None)
    _, loss_dict, _ = pipeline.get_train_loss_dict(step)
    loss = functools.reduce(torch.add, loss_dict.values())
    grad_scaler.scale(loss).backward()              # GradScaler(enabled=mixed_precision): disabled for `neusky` (neusky_config.py:38)
    for optimizer: grad_scaler.step(optimizer)      # Optimizers.optimizer_scaler_step_all: torch.optim.Adam(eps=1e-15) per group, :216-237
    grad_scaler.update()
    for scheduler: scheduler.step()

-- driving NeuSkyPipeline with nothing from neusky_amd.engine.  The gradient exchange (world_size 2, gloo, both ranks on cuda:0; or a
one-rank RCCL group) and the graph replay live behind get_train_loss_dict.

    python tests/trainer_loop_worker.py <rank> <world> <port> <out.npz> <backend> <steps> <eager_steps>

world = 1, backend "none": the single-process reference on the concatenated batch."""
import functools
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


STEP0 = 10_000


def torch_optimizers(pipe, step0=STEP0, fused=False):
    """what nerfstudio's Optimizers builds from neusky_config.py:216-237: one Adam per parameter group + its scheduler as a LambdaLR
    (fused: neusky_amd.optimizers.SlabAdam, what plugin.SlabAdamOptimizerConfig hands the trainer; else torch.optim.Adam)"""
    import torch
    from neusky_amd.engine import ExponentialDecaySchedulerConfig, neusky_optimizers
    from neusky_amd.optimizers import SlabAdam
    cfg = neusky_optimizers()
    opts, scheds = {}, {}
    Adam = SlabAdam if fused else torch.optim.Adam
    for name, params in pipe.get_param_groups().items():
        oc, sc = cfg[name]["optimizer"], cfg[name]["scheduler"]
        opts[name] = Adam([p for p in params if p.requires_grad], lr=oc.lr, eps=oc.eps, betas=oc.betas)
        if isinstance(sc, ExponentialDecaySchedulerConfig):
            sc.lr_init = oc.lr
        scheds[name] = torch.optim.lr_scheduler.LambdaLR(opts[name], lr_lambda=lambda e, f=sc.factor: f(step0 + e))  # (the trainer resumes at step0)
    return opts, scheds


_SCALER = []


def nerfstudio_train_iteration(pipe, opts, scheds, step, **inject):
    """nerfstudio Trainer.train_iteration, statement for statement (its GradScaler is disabled unless mixed_precision is set)"""
    import torch
    if not _SCALER:
        _SCALER.append(torch.amp.GradScaler("cuda", enabled=False))
    grad_scaler = _SCALER[0]
    for o in opts.values():
        o.zero_grad()
    _, loss_dict, _ = pipe.get_train_loss_dict(step, **inject)
    loss = functools.reduce(torch.add, loss_dict.values())
    grad_scaler.scale(loss).backward()
    for o in opts.values():
        grad_scaler.step(o)
    grad_scaler.update()
    for s in scheds.values():
        s.step()
    return loss.detach()


def main():
    rank, world, port, out, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    steps, eager_steps = int(sys.argv[6]), int(sys.argv[7])
    import numpy as np
    import torch
    import torch.distributed as dist
    from two_rank_worker import build, shard
    dev = "cuda:0"
    if backend != "none":
        import datetime
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    pipe, rb, batch, rnd = build(dev, world, rank)
    rbs, bs, rs = shard(rb, batch, rnd, world, rank, dev)
    pipe.config.graph_replay, pipe.config.graph_replay_warmup = True, eager_steps
    if backend == "nccl" and world == 1:
        pipe.exchange_with_one_rank()  # the RCCL exchange on the hardware there is: one rank
    opts, scheds = torch_optimizers(pipe, fused=bool(os.environ.get("NSKY_TEST_SLAB_ADAM")))
    before = {n: p.detach().clone() for n, p in pipe.named_parameters() if p.requires_grad}
    losses, grads_eager = [], None
    for i in range(steps):
        losses.append(float(nerfstudio_train_iteration(pipe, opts, scheds, STEP0 + i, ray_bundle=rbs, batch=bs, randoms=rs)))
        if i == 0:  # the exchanged gradient of the first (eager) pass, as the optimizers saw it
            grads_eager = {n: p.grad.detach().cpu().numpy().copy() for n, p in pipe.named_parameters() if p.requires_grad and p.grad is not None}
    torch.cuda.synchronize()
    assert (pipe._train_graph is not None) == (steps > eager_steps), "graph replay did not engage when it should"
    grads_last = {n: p.grad.detach().cpu().numpy().copy() for n, p in pipe.named_parameters() if p.requires_grad and p.grad is not None}
    slab = pipe.gradient_slab()
    in_slab = all(p.grad is None or (p.grad.data_ptr() >= slab.flat.data_ptr() and p.grad.data_ptr() < slab.flat.data_ptr() + 4 * slab.flat.numel())
                  for p, _ in slab.views) if (world > 1 or steps > eager_steps or backend == "nccl") else True
    params = {n: p.detach().cpu().numpy().copy() for n, p in pipe.named_parameters() if p.requires_grad}
    moved = sum(int(not torch.equal(p.detach(), before[n])) for n, p in pipe.named_parameters() if p.requires_grad)
    if world > 1:  # the replicas took the same steps
        digest = torch.tensor([float(np.abs(v.astype(np.float64)).sum()) for v in params.values()], dtype=torch.float64)
        both = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        assert all(torch.equal(b, both[0]) for b in both), "the replicas diverged under the trainer loop"
    if rank == 0:
        np.savez(out, losses=np.array(losses), moved=moved, in_slab=in_slab,
                 **{"g0:" + k: v for k, v in grads_eager.items()}, **{"g:" + k: v for k, v in grads_last.items()},
                 **{"p:" + k: v for k, v in params.items()}, **{"b:" + k: v.cpu().numpy() for k, v in before.items()})
    if backend != "none":
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
