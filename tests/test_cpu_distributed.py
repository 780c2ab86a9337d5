"""world_size-2 gloo test of the ray-sharded data-parallel path: replicas start identical (broadcast), the flat
gradient all-reduce averages the per-rank gradients, parameters without a gradient are zero-filled
(find_unused_parameters semantics), and the averaged gradient equals the gradient of the concatenated batch."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _RendezvousError(RuntimeError):
    pass


def _init(rank, world, port, out_q) -> bool:
    """rendezvous; a failure HERE (the port found free a moment ago was taken) is reported as such, so that only it is retried"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    try:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        return True
    except Exception as exc:  # noqa: BLE001
        out_q.put(("rendezvous_error", rank, f"{type(exc).__name__}: {exc}"))
        return False


def _collect(q, procs, timeout):
    res = []
    try:
        for _ in procs:
            item = q.get(timeout=timeout)
            if item and item[0] == "rendezvous_error":
                raise _RendezvousError(item[2])
            res.append(item)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return sorted(res, key=lambda t: t[0])


def _worker(rank, world, port, out_q):
    if not _init(rank, world, port, out_q):
        return
    from neusky_amd.distributed import ReplicaSync
    from neusky_amd.engine import AdamOptimizerConfig, Optimizers
    torch.manual_seed(100 + rank)  # deliberately different initial replicas
    mod = torch.nn.Module()
    mod.w = torch.nn.Parameter(torch.randn(5, 3))
    mod.b = torch.nn.Parameter(torch.randn(5))
    mod.unused = torch.nn.Parameter(torch.randn(4))  # receives no gradient on any rank
    w, b, unused = mod.w, mod.b, mod.unused
    sync = ReplicaSync(mod, world)
    sync.broadcast_parameters()
    sync.barrier()
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 3, generator=g); Y = torch.randn(8, 5, generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]  # each rank owns its rays
    # reference: one process, concatenated batch (batch-mean losses => average, not sum, of the shard gradients)
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ((X @ w2.T + b2 - Y) ** 2).mean().backward()
    # the engine's form (bench.py / train loop): every optimizer group's gradients live in ONE slab that is all-reduced once;
    # a parameter that receives no gradient keeps its zero-filled slot (find_unused_parameters semantics, neusky_pipeline.py:199)
    w_start = w.detach().clone()
    opt = Optimizers({"fields": {"optimizer": AdamOptimizerConfig(), "scheduler": None},
                      "ddf_field": {"optimizer": AdamOptimizerConfig(), "scheduler": None}},
                     {"fields": [w, unused], "ddf_field": [b]}, world_size=world)
    opt.zero_grad_all()
    ((xs @ w.T + b - ys) ** 2).mean().backward()
    opt.all_reduce_gradients()
    slab_ok = w.grad.data_ptr() == opt.flat_g.data_ptr() and torch.equal(w.detach(), w_start)
    # numpy, not tensors: a tensor crosses the queue as a shared-memory handle that dies with this process
    out_q.put((rank,) + tuple(t.detach().numpy().copy() for t in (w, w.grad, b.grad, unused.grad, w2.grad, b2.grad)) + (bool(slab_ok),))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    return _collect(q, procs, 120)


def _run_with_one_rendezvous_retry(fn):
    """the rendezvous port found free a moment ago can be taken by the time rank 0 binds it: ONE retry, for exactly that failure
    (reported by the worker itself); a crashed rank, a hang or a wrong result is never retried"""
    try:
        return fn()
    except _RendezvousError:
        return fn()


def test_two_rank_gradient_allreduce_equals_single_process():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")  # two freshly spawned interpreters must not race on __pycache__
    res = _run_with_one_rendezvous_retry(_run_two_ranks)
    (_, w0, gw0, gb0, gu0, rw0, rb0, ok0), (_, w1, gw1, gb1, gu1, _, _, ok1) = [
        tuple(torch.from_numpy(v) if hasattr(v, "dtype") else v for v in item) for item in res]
    assert ok0 and ok1, "engine.Optimizers: the gradients are not views of the one slab"
    assert torch.equal(w0, w1), "replicas differ after the parameter broadcast"
    assert torch.allclose(gw0, gw1) and torch.allclose(gb0, gb1), "ranks disagree after the all-reduce"
    assert torch.allclose(gw0, rw0, atol=1e-6) and torch.allclose(gb0, rb0, atol=1e-6), "N-GPU gradient != 1-GPU gradient"
    assert torch.equal(gu0, torch.zeros(4)) and torch.equal(gu1, torch.zeros(4))


# ---------------------------------------------------------------------------------------------------------------------
# the train loop's reduce path with the REAL parameter-group layout
def _pipeline_worker(rank, world, port, out_q):
    if not _init(rank, world, port, out_q):
        return
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util_step import small_pipeline_config
    from neusky_amd.engine import Optimizers, neusky_optimizers
    torch.manual_seed(1000 + rank)  # every rank draws its own initial weights, frozen RENI decoder included
    pipe = small_pipeline_config(R=8, images=3).setup(device="cpu", world_size=world, local_rank=rank)  # broadcasts the full module state
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups(), world_size=world)
    # layout facts: one slab, groups back to back in the optimizer-config order, every parameter on a 16-byte boundary
    off, layout_ok, names = 0, True, []
    for g in opt.groups:
        names.append(g.name)
        layout_ok &= g.flat_g.data_ptr() == opt.flat_g.data_ptr() + 4 * off
        o = 0
        for p in g.params:
            layout_ok &= p.grad.data_ptr() == g.flat_g.data_ptr() + 4 * o and p.data.data_ptr() == g.flat_p.data_ptr() + 4 * o and o % 4 == 0
            o += (p.numel() + 3) // 4 * 4
        layout_ok &= o == g.numel
        off += g.numel
    layout_ok &= off == opt.flat_g.numel()
    frozen = [p for p in pipe.parameters() if not p.requires_grad]
    in_slab = {id(p) for g in opt.groups for p in g.params}
    frozen_ok = len(frozen) > 0 and all(id(p) not in in_slab for p in frozen)
    # stand-in backward: rank-dependent gradients for every parameter of every group except a few that stay untouched (a
    # head that received no gradient this step: find_unused_parameters semantics = the zero fill of zero_grad_all survives)
    opt.zero_grad_all()
    skipped = []
    for gi, g in enumerate(opt.groups):
        for pi, p in enumerate(g.params):
            if (gi + pi) % 5 == 4:
                skipped.append((gi, pi))
                continue
            gen = torch.Generator().manual_seed(31 * gi + pi)  # same base on both ranks
            base = torch.randn(p.shape, generator=gen)
            p.grad.copy_(base * (rank + 1))  # mean over ranks = 1.5 * base
    def fill():
        opt.zero_grad_all()
        for gi, g in enumerate(opt.groups):
            for pi, p in enumerate(g.params):
                if (gi, pi) in skipped:
                    continue
                gen = torch.Generator().manual_seed(31 * gi + pi)
                p.grad.copy_(torch.randn(p.shape, generator=gen) * (rank + 1))

    def check():
        ok = True
        for gi, g in enumerate(opt.groups):
            for pi, p in enumerate(g.params):
                gen = torch.Generator().manual_seed(31 * gi + pi)
                base = torch.randn(p.shape, generator=gen)
                want = torch.zeros_like(base) if (gi, pi) in skipped else 1.5 * base
                ok &= bool(torch.allclose(p.grad, want, atol=1e-6))
        return ok

    opt.all_reduce_gradients()
    reduce_ok = check()
    # the bucketed exchange: asynchronous all-reduces of the two slab ranges (DDF / scalar / latent groups first, then field + proposal),
    # each group usable once ITS bucket has been waited for; same result as the single message
    fill()
    opt.all_reduce_gradients(bucketed=True)
    buckets = [[g.name for g in gs] for gs, _ in opt._buckets()]
    reduce_ok &= buckets == [["illumination_field", "visibility_sigmoid", "ddf_field"], ["proposal_networks", "fields"]] and len(opt._pending) == 2
    for g in opt.groups:
        opt._wait_bucket_of(g)
    reduce_ok &= len(opt._pending) == 0 and check()
    state = torch.cat([t.detach().reshape(-1).float() for t in list(pipe.parameters()) + list(pipe.buffers())])
    out_q.put((rank, names, bool(layout_ok), bool(frozen_ok), bool(reduce_ok), len(skipped), float(state.double().sum()),
               float(state.double().abs().sum()), int(opt.flat_g.numel())))
    dist.barrier()
    dist.destroy_process_group()


def _run_pipeline_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    return _collect(q, procs, 300)


def test_two_rank_pipeline_groups_slab_and_replicas():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    r0, r1 = _run_with_one_rendezvous_retry(_run_pipeline_ranks)
    assert r0[1] == r1[1] == ["proposal_networks", "fields", "illumination_field", "visibility_sigmoid", "ddf_field"]  # neusky_config.py:216-237
    for r in (r0, r1):
        assert r[2], "gradient slab layout: groups / parameters are not contiguous 16-byte aligned views of ONE slab"
        assert r[3], "the frozen RENI decoder must stay outside the gradient slab"
        assert r[4], "all-reduced slab != mean of the rank gradients (or an untouched slot was not zero)"
        assert r[5] > 0
    assert r0[6] == r1[6] and r0[7] == r1[7], "replicas differ after the state broadcast (frozen parameters / buffers included)"
    assert r0[8] == r1[8]


# ---------------------------------------------------------------------------------------------------------------------
# the RCCL branch of the exchange (engine.Optimizers.all_reduce_gradients): no process group here, the collectives are recorded
def test_nccl_branch_of_the_gradient_exchange(monkeypatch):
    """with backend "nccl" the slab is all-reduced in place with ReduceOp.AVG -- once, or once per bucket asynchronously -- and the
    optimizer step of a group waits for the bucket that holds it"""
    import neusky_amd.engine as E
    params = {k: [torch.nn.Parameter(torch.randn(n))] for k, n in (("proposal_networks", 8), ("fields", 12), ("illumination_field", 4),
                                                                     ("visibility_sigmoid", 1), ("ddf_field", 16))}
    opt = E.Optimizers(E.neusky_optimizers(), params, world_size=8)
    calls, waited, stepped = [], [], []

    class Work:
        def __init__(self, i):
            self.i = i

        def wait(self):
            waited.append(self.i)

    def fake_all_reduce(t, op=None, async_op=False):
        calls.append((t.data_ptr(), t.numel(), op, async_op))
        return Work(len(calls) - 1) if async_op else None

    monkeypatch.setattr(E.dist, "get_backend", lambda: "nccl")
    monkeypatch.setattr(E.dist, "all_reduce", fake_all_reduce)
    monkeypatch.setattr(E.hip, "adam_step", lambda p, g, *a: stepped.append((g.data_ptr(), list(waited))))
    opt.all_reduce_gradients()
    assert calls == [(opt.flat_g.data_ptr(), opt.flat_g.numel(), E.dist.ReduceOp.AVG, False)]
    calls.clear()
    opt.all_reduce_gradients(bucketed=True)
    n_a = sum(g.numel for g in opt.groups if g.name in ("illumination_field", "visibility_sigmoid", "ddf_field"))
    n_b = sum(g.numel for g in opt.groups if g.name in ("proposal_networks", "fields"))
    ptr = {g.name: g.flat_g.data_ptr() for g in opt.groups}
    assert calls == [(ptr["illumination_field"], n_a, E.dist.ReduceOp.AVG, True), (ptr["proposal_networks"], n_b, E.dist.ReduceOp.AVG, True)]
    opt.optimizer_scheduler_step_all(10)
    # Adam launches in bucket order; the first bucket's groups only waited for collective 0
    assert [s[0] for s in stepped] == [ptr[k] for k in ("illumination_field", "visibility_sigmoid", "ddf_field", "proposal_networks", "fields")]
    assert stepped[0][1] == [0] and stepped[2][1] == [0] and stepped[3][1] == [0, 1] and opt._pending == []
    # a single process never touches the process group
    calls.clear()
    solo = E.Optimizers(E.neusky_optimizers(), {k: [torch.nn.Parameter(torch.randn(4))] for k in params}, world_size=1)
    solo.all_reduce_gradients()
    assert calls == []


# ---------------------------------------------------------------------------------------------------------------------
# camera-sharded illumination decode: the all-gather of the colours and the reduce-scatter of their gradient (distributed.CameraAllGather)
def _camera_gather_worker(rank, world, port, out_q):
    if not _init(rank, world, port, out_q):
        return
    from neusky_amd.distributed import CameraAllGather
    U, D = 5, 4  # cameras 0, 2, 4 on rank 0 (three rows), 1, 3 on rank 1 (two rows + one row of padding in the message)
    own = torch.arange(rank, U, world)
    f = lambda cams: (cams[:, None, None] * 100 + torch.arange(D)[None, :, None] * 10 + torch.arange(3)[None, None, :]).double()  # noqa: E731
    local = f(own).clone().requires_grad_(True)
    cols = CameraAllGather.apply(local, U, rank, world)
    fwd_ok = bool(torch.equal(cols, f(torch.arange(U))))
    w = torch.randn(U, D, 3, generator=torch.Generator().manual_seed(5 + rank), dtype=torch.float64)  # this rank's rays' gradient w.r.t. ALL cameras
    (cols * w).sum().backward()
    w_all = sum(torch.randn(U, D, 3, generator=torch.Generator().manual_seed(5 + r), dtype=torch.float64) for r in range(world))
    bwd_ok = bool(torch.allclose(local.grad, w_all[own])) and local.grad.shape == local.shape
    out_q.put((rank, fwd_ok, bwd_ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_camera_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_camera_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    return _collect(q, procs, 120)


def test_camera_all_gather_and_reduce_scatter_two_ranks():
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    res = _run_with_one_rendezvous_retry(_run_camera_gather)
    assert all(r[1] for r in res), "all-gathered colours are not in camera order"
    assert all(r[2] for r in res), "the owner's gradient is not the sum over ranks of the gradient w.r.t. its cameras' colours"
