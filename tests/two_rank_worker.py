"""One rank of tests/test_gpu_two_ranks.py: a fresh process on cuda:0, `gloo` rendezvous, the REAL pipeline.

    python tests/two_rank_worker.py <rank> <world> <port> <out.npz>

Every rank builds the same seeded pipeline (world_size = 2: the constructor broadcasts rank 0's state), takes its half of one
seeded 32-ray batch with the matching rows of the injected random draws, runs one eager iteration (zero-grad, forward, losses,
backward, gradient all-reduce) and then the same iteration as a HIP-graph replay followed by the all-reduce, and writes the
all-reduced gradient of every parameter.  world = 1: the single-process reference on the concatenated batch."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

R_TOTAL = 32


def build(dev, world, rank):
    import torch
    from util_step import make_randoms, randomise, small_pipeline_config
    torch.manual_seed(0)
    cfg = small_pipeline_config(R=R_TOTAL // world, num_prop=(24, 12), S=8, D=24, images=5, vmf=(2, 8), sky=8)
    cfg.shard_illumination_decode = bool(os.environ.get("NSKY_TEST_SHARD_ILLUMINATION")) and world > 1  # (tests/test_gpu_two_ranks.py)
    pipe = cfg.setup(device=dev, world_size=world, local_rank=rank)
    pipe.train()
    randomise(pipe)
    # ONE seeded batch for every process: a datamanager of its own with rank 0's generator
    from neusky_amd.data.synthetic_datamanager import SyntheticDataManagerConfig
    dm = SyntheticDataManagerConfig(num_train_images=5, num_eval_images=2, train_num_rays_per_batch=R_TOTAL).setup(device=dev)
    rb, batch = dm.next_train(0)
    pipe.datamanager = dm  # make_randoms draws the sky bundle from it
    rnd = make_randoms(pipe, R_TOTAL)
    return pipe, rb, batch, rnd


def shard(rb, batch, rnd, world, rank, dev):
    from neusky_amd.cameras.rays import RayBundle
    from util_step import randoms_to
    n = R_TOTAL // world
    sl = slice(rank * n, (rank + 1) * n)
    c = lambda t: t[sl].contiguous()  # noqa: E731
    rbs = RayBundle(origins=c(rb.origins), directions=c(rb.directions), pixel_area=c(rb.pixel_area), camera_indices=c(rb.camera_indices),
                    metadata={k: c(v) for k, v in rb.metadata.items()})
    bs = {"image": c(batch["image"]), "mask": c(batch["mask"])}
    r = dict(rnd)
    r["jitters"] = [j[sl].contiguous() for j in rnd["jitters"]]  # per-ray draws follow their rays; everything else is shared
    return rbs, bs, randoms_to(r, dev)


def named_grads(pipe):
    return {n: p.grad.detach().cpu().numpy().copy() for n, p in pipe.named_parameters() if p.requires_grad and p.grad is not None}


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    import numpy as np
    import torch
    import torch.distributed as dist
    dev = "cuda:0"
    if world > 1:
        import datetime
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    from neusky_amd.model_components.losses import total_loss
    pipe, rb, batch, rnd = build(dev, world, rank)
    rbs, bs, rs = shard(rb, batch, rnd, world, rank, dev)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups(), world_size=world)
    # ---- eager iteration + all-reduce (no optimizer step: the graph run below starts from the same parameters)
    opt.zero_grad_all()
    _, ld, _ = pipe.get_train_loss_dict(10_000, ray_bundle=rbs, batch=bs, randoms=rs)
    loss = total_loss(ld)
    loss.backward()
    opt.collect_grads()
    opt.all_reduce_gradients()
    torch.cuda.synchronize()
    eager = named_grads(pipe)
    slab_eager = opt.flat_g.detach().cpu().numpy().copy()
    # nothing of the eager iteration's autograd graph may outlive it: a live loss tensor keeps the parameters' AccumulateGrad nodes,
    # which remember the (legacy) stream they were made on and would run on it inside the capture below
    loss_eager = float(loss)
    del loss, ld
    if getattr(pipe.model, "illumination_shard", None) is not None:  # (the colours' all-gather cannot be captured under gloo: eager only)
        if rank == 0:
            np.savez(out, loss=loss_eager, slab_eager=slab_eager, **{"g:" + k: v for k, v in eager.items()})
        dist.barrier()
        dist.destroy_process_group()
        return
    # ---- the same iteration captured in a HIP graph, replayed, all-reduced, Adam-stepped
    before = {n: p.detach().clone() for n, p in pipe.named_parameters() if p.requires_grad}
    stepper = GraphedTrainStep(pipe, opt, rbs, bs, warmup=1, start_step=10_000, randoms=rs)
    gl, _, _ = stepper.step(10_000, rbs, bs, rs["sky_ray_bundle"])  # (the sky rays of the eager iteration: the stepper drew its own)
    torch.cuda.synchronize()
    slab_graph = opt.flat_g.detach().cpu().numpy().copy()
    moved = sum(int(not torch.equal(p.detach(), before[n])) for n, p in pipe.named_parameters() if p.requires_grad)
    if world > 1:
        sums = torch.tensor([float(np.abs(slab_eager).sum()), float(np.abs(slab_graph).sum())], dtype=torch.float64)
        both = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(both, sums)
        assert all(torch.equal(b, both[0]) for b in both), "ranks hold different slabs after the all-reduce"
    if rank == 0:
        np.savez(out, loss=loss_eager, graph_loss=float(gl), slab_eager=slab_eager, slab_graph=slab_graph, moved=moved,
                 **{"g:" + k: v for k, v in eager.items()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
