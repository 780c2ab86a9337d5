"""Ray-sharded data parallelism: one process per GPU, replicated parameters, ONE flat gradient
all-reduce (mean) per step over RCCL/xGMI (`torch.distributed` backend "nccl"; "gloo" in CPU tests).

Replaces `DDP(self._model, device_ids=[local_rank], find_unused_parameters=True)` + `dist.barrier`
at neusky/pipelines/neusky_pipeline.py:198-200: `ReplicaSync` (state broadcast, barrier), `GradientSlab` (the reducer: layout, zero
fill, collection, all-reduce, owned by NeuSkyPipeline and exchanged at the end of the trainer's backward pass), `CameraAllGather`
(the opt-in camera-sharded illumination decode).  `find_unused_parameters=True` semantics are kept by
zero-filling the slots of parameters that received no gradient (frozen RENI decoder, unused heads), so
ranks never disagree on the message.  The payload is dominated by the two 2^19 x 16 hash tables
(2 x 48.8 MB fp32); xGMI ring all-reduce of ~110 MB costs about a millisecond against a >15 ms step,
so a single bucket after backward is used rather than overlapped buckets.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def broadcast_module_state(module: torch.nn.Module, src: int = 0) -> int:
    """Make every replica identical to rank `src`: EVERY parameter (frozen ones included - the RENI decoder is frozen but
    drawn from each process's own RNG) and every buffer, like the DDP wrapper's initial state broadcast
    (neusky/pipelines/neusky_pipeline.py:198-199).  Returns the number of tensors sent."""
    n = 0
    seen = set()
    for t in list(module.parameters()) + list(module.buffers()):
        if id(t) in seen:
            continue
        seen.add(id(t))
        _broadcast(t.data, src)
        n += 1
    return n


def _broadcast(t: torch.Tensor, src: int) -> None:
    """RCCL broadcasts device tensors in place; a tensor on the other side of the backend is staged: a device tensor through the host
    under gloo (two ranks sharing one GPU in tests), a host tensor (the pipeline's `num_*_data` buffers, host-resident latents) through
    the current device under RCCL -- which has no CPU backend ("No backend type associated with device type cpu")."""
    backend = dist.get_backend()
    if t.is_cuda and backend != "nccl":
        host = t.cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    elif not t.is_cuda and backend == "nccl":
        dev = t.to(torch.device("cuda", torch.cuda.current_device()))
        dist.broadcast(dev, src=src)
        t.copy_(dev)
    else:
        dist.broadcast(t, src=src)


class ReplicaSync:
    """The DDP wrapper's start-up half: the initial state broadcast and the barrier of neusky_pipeline.py:198-200.  (Its per-step
    half -- the gradient reducer -- is `GradientSlab` below, owned by the pipeline.)"""

    def __init__(self, module: torch.nn.Module, world_size: int):
        self.module = module
        self.world_size = world_size

    def broadcast_parameters(self, src: int = 0) -> int:
        return broadcast_module_state(self.module, src)

    def barrier(self) -> None:
        dist.barrier()


# ---------------------------------------------------------------------------------------------------------------------
# The gradient slab: what DDP's reducer is to the reference (neusky_pipeline.py:198-199), owned by the PIPELINE so that any trainer
# that drives `get_train_loss_dict` -> `backward` (nerfstudio's Trainer with torch.optim.Adam, or neusky_amd.engine) sees synchronised
# gradients in `p.grad`.
GROUP_ORDER = ("proposal_networks", "fields", "illumination_field", "visibility_sigmoid", "ddf_field")  # neusky_config.py:216-237
# buckets of the exchange, in the order the backward pass finishes them: the DDF group (its chain backward and hash-table scatter run
# first) with the scalar / latent groups, then the field and proposal groups (last to finish).  Segments of one bucket are adjacent in
# the slab when the groups are in GROUP_ORDER; otherwise the bucket falls back to per-group messages.
COMM_BUCKETS = (("ddf_field", "visibility_sigmoid", "illumination_field"), ("fields", "proposal_networks"))


def slab_of(params) -> "GradientSlab | None":
    """the live GradientSlab the first of `params` belongs to (parameters carry a weak reference), or None"""
    for p in params:
        ref = getattr(p, "_nsky_slab", None)
        slab = ref() if ref is not None else None
        return slab
    return None


class GradientSlab:
    """ONE flat fp32 buffer holding the gradient of every trainable parameter of the pipeline's groups, group after group, every
    parameter on a 16-byte boundary.  `p.grad` of each parameter is a view of it, so
      * the HIP backward kernels accumulate weight / bias / hash-table gradients straight into it (ops.register_grad_sink),
      * the multi-GPU exchange is ONE all-reduce (mean) of ~110 MB over RCCL / xGMI instead of one message per tensor,
      * an optimizer -- torch.optim.Adam on the parameters, or the fused `nsky_adam_step` of neusky_amd.engine -- reads it in place.
    A parameter that received no gradient in a pass keeps its zero-filled slot, so every rank sends the same message whatever was used
    (`find_unused_parameters=True`, neusky_pipeline.py:199)."""

    def __init__(self, param_groups, world_size: int = 1, order=None):
        import weakref
        order = list(GROUP_ORDER if order is None else order)
        names = [k for k in order if k in param_groups] + [k for k in param_groups if k not in order]
        self.world_size = world_size
        self.groups = []  # [(name, [params], numel)]
        for k in names:
            ps = [p for p in param_groups[k] if p.requires_grad]
            if ps:
                self.groups.append((k, ps, sum((p.numel() + 3) // 4 * 4 for p in ps)))
        if not self.groups:
            raise ValueError("no trainable parameter in any group")
        dev = self.groups[0][1][0].device
        self.flat = torch.zeros(sum(n for _, _, n in self.groups), device=dev)
        self.group_views, self.views = {}, []  # name -> slab range; [(p, view)]
        off = 0
        me = weakref.ref(self)
        for name, ps, n in self.groups:
            self.group_views[name] = self.flat[off:off + n]
            o = off
            for p in ps:
                k = p.numel()
                view = self.flat[o:o + k].view_as(p)
                p.grad = view
                p._nsky_grad_sink = True  # custom backward passes may accumulate into p.grad directly (zeroed before every pass)
                p._nsky_slab = me
                self.views.append((p, view))
                o += (k + 3) // 4 * 4
            off += n
        self.rebind()
        self._backward_seen = False
        self._clean = True  # every slot no gradient was written to since the last fill is zero
        self.used = [True] * len(self.views)
        self._pending = []
        self.force_exchange = False  # run the collective with one rank too (a one-rank RCCL group: bench.py's self-check, tests)
        self.exchanged = False  # the pipeline's end-of-pass hook has already exchanged this pass's gradients
        # The ONE host seam: a slab in host memory exists only in the world-size-2 `gloo` tests of the exchange logic
        # (tests/test_cpu_distributed.py: layout, buckets, zero-fill semantics, mean) -- there autograd accumulates into the slab views
        # and nothing is gathered; every training path has the slab in HBM.
        self.host = dev.type == "cpu"

    def rebind(self) -> None:
        """(again) after the parameters were re-homed (engine._Group moves `p.data` into its parameter slab): the sinks are keyed by address"""
        from . import ops
        for p, _ in self.views:
            ops.register_grad_sink(p)

    # ------------------------------------------------------------------ one pass
    def zero_all(self) -> None:
        """engine form (and the inside of a captured step): zero the slab; parameters whose gradient no kernel writes into the slab itself
        (everything behind weight norm / padding / plain torch ops) start the backward with an undefined .grad: autograd's AccumulateGrad
        then keeps the incoming tensor instead of launching one add kernel per parameter, and collect() moves all of them into the slab
        with one launch.  Between this call and collect() the .grad of such a parameter is None or an autograd-owned temporary."""
        self.flat.zero_()
        self.exchanged, self._clean = False, True
        if self._backward_seen and not self.host:
            for p, _ in self.views:
                if not getattr(p, "_nsky_sunk", False):
                    p.grad = None

    def begin_pass(self) -> None:
        """trainer form: called by the pipeline at the top of get_train_loss_dict, i.e. after the trainer's own zero_grad (torch's sets
        .grad = None; set_to_none=False zeroes the views in place and keeps them).  Only the parameters the kernels accumulate into IN
        PLACE need anything here: a zeroed slot and their view back as .grad.  Every other slot is either overwritten by collect() or
        zeroed there (no gradient this pass)."""
        self.exchanged = False
        if self.host:
            return
        none = [p.grad is None for p, _ in self.views]
        if all(none):
            self.flat.zero_()  # (the usual case: ONE fill)
            self._clean = True
        for (p, view), n in zip(self.views, none):
            if n and getattr(p, "_nsky_sunk", False):
                if not self._clean:
                    view.zero_()
                p.grad = view

    def collect(self, attach_unused: bool = True) -> None:
        """after backward: every gradient autograd left outside the slab is moved into its view (ONE launch) and .grad is the view again.
        attach_unused=False (trainer form): a parameter that received no gradient keeps .grad = None, as under DDP when no rank used it
        (torch optimizers then skip it) -- which parameters a step uses depends on the config and the step only, never on the rank's rays,
        so ranks agree; its slot is zero in the message either way.  `self.used`: which parameters hold a gradient of this pass."""
        self._backward_seen = True
        if self.host:
            return
        from . import hip
        pairs, used = [], []
        for p, view in self.views:
            g = p.grad
            used.append(g is not None)
            if g is None:
                if not self._clean:
                    view.zero_()
                if attach_unused:
                    p.grad = view
                continue
            if g.data_ptr() != view.data_ptr():
                pairs.append((g if g.is_contiguous() else g.contiguous(), view))
            p.grad = view
        self.used = used
        self._clean = False  # (until the next fill)
        if pairs:
            hip.gather_segments(pairs)

    # ------------------------------------------------------------------ exchange
    def buckets(self):
        """[(group names, slab view)]: the contiguous slab range covered by the groups of each bucket that exist"""
        out, offs, off = [], {}, 0
        for name, _, n in self.groups:
            offs[name] = (off, off + n)
            off += n
        done = set()
        for names in COMM_BUCKETS:
            gs = [name for name, _, _ in self.groups if name in names]
            spans = sorted(offs[g] for g in gs)
            if gs and all(a[1] == b[0] for a, b in zip(spans, spans[1:])):
                out.append((gs, self.flat[spans[0][0]:spans[-1][1]]))
            else:
                out += [([g], self.group_views[g]) for g in gs]
            done |= set(gs)
        out += [([name], self.group_views[name]) for name, _, _ in self.groups if name not in done]
        return out

    def all_reduce(self, bucketed: bool = False) -> None:
        """mean over ranks, over RCCL / xGMI: ONE all-reduce of the whole slab (~110 MB, 0.5-1.3 ms on 8 x MI355X against a 20 ms
        step), or -- bucketed -- one asynchronous all-reduce per bucket, issued back to back on RCCL's own stream;
        wait(group) then makes the consumer of a group wait for its own bucket only, so the second bucket's exchange overlaps the first
        bucket's optimizer launches."""
        self._pending = []
        if self.world_size <= 1 and not self.force_exchange:
            return
        backend = dist.get_backend()
        if backend == "nccl":
            if bucketed:
                self._pending = [(gs, dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=True)) for gs, view in self.buckets()]
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG)
        elif not self.host:  # gloo with device gradients (two ranks sharing one GPU in tests): staged through the host
            host = self.flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat.copy_(host.div_(self.world_size))
        elif bucketed:  # gloo (CPU tests) has no AVG: asynchronous SUMs, the division happens when the bucket is waited for
            self._pending = [(gs, dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view) for gs, view in self.buckets()]
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(self.world_size)

    def wait(self, group: "str | None" = None) -> None:
        """the current stream (RCCL) / the host (gloo) waits for the bucket holding `group`; None: for all of them"""
        for item in list(self._pending):
            if group is None or group in item[0]:
                item[1].wait()
                if len(item) == 3:
                    item[2].div_(self.world_size)
                self._pending.remove(item)


# ---------------------------------------------------------------------------------------------------------------------
# Camera-sharded illumination decode (the one place the path shards beyond rays; neusky_model.py:445-551).  In a training step EVERY
# training camera's environment light is decoded at the step's D directions (static shapes: U x D rows whatever the batch holds) --
# 153 600 rows, ~3 ms of the 20 ms step that every rank would repeat.  With `NeuSkyPipelineConfig.shard_illumination_decode` rank r
# decodes cameras r::N and the [U, D, 3] colours are all-gathered (1.8 MB); backward, each rank's gradient w.r.t. ALL cameras' colours
# (from its own rays) is reduce-scattered (sum) to the owners, which backpropagate it through their share of the decode.  The owner's
# latent gradient is then the SUM over ranks of what plain data parallelism would have spread over them, every other rank's is zero,
# and the slab's all-reduce(mean) gives sum / N on every rank: exactly the mean the unsharded replicas get.  All ranks must use the same
# direction set in a step (the sampler's generator is seeded rank-free in this mode).
def _all_gather_rows(local: torch.Tensor) -> torch.Tensor:
    """[n, ...] (same shape on every rank) -> [N, n, ...]"""
    n = dist.get_world_size()
    if dist.get_backend() == "nccl" or not local.is_cuda:
        if dist.get_backend() == "nccl":
            out = torch.empty((n,) + tuple(local.shape), device=local.device, dtype=local.dtype)
            dist.all_gather_into_tensor(out, local.contiguous())
            return out
        outs = [torch.empty_like(local) for _ in range(n)]
        dist.all_gather(outs, local.contiguous())
        return torch.stack(outs)
    host = local.detach().cpu()  # gloo with device tensors (two ranks sharing one GPU in tests): staged through the host
    outs = [torch.empty_like(host) for _ in range(n)]
    dist.all_gather(outs, host)
    return torch.stack(outs).to(local.device)


def _reduce_scatter_rows(full: torch.Tensor) -> torch.Tensor:
    """[N, n, ...] on every rank -> sum over ranks of full[this rank]: [n, ...]"""
    if dist.get_backend() == "nccl":
        out = torch.empty(tuple(full.shape[1:]), device=full.device, dtype=full.dtype)
        dist.reduce_scatter_tensor(out, full.contiguous(), op=dist.ReduceOp.SUM)
        return out
    host = full.detach().cpu().contiguous()  # gloo has no reduce-scatter: all-reduce, keep the own block
    dist.all_reduce(host, op=dist.ReduceOp.SUM)
    return host[dist.get_rank()].to(full.device)


class CameraAllGather(torch.autograd.Function):
    """colours of the cameras rank r decoded (cameras r, r + N, r + 2N, ...: [ceil((U - r) / N), D, 3]) -> colours of all U cameras
    on every rank; backward: reduce-scatter(sum) of the gradient w.r.t. all cameras' colours"""

    @staticmethod
    def forward(ctx, local, U, rank, world):
        n_pad = (U + world - 1) // world
        n_loc = local.shape[0]
        if n_loc < n_pad:
            local = torch.cat([local, local.new_zeros((n_pad - n_loc,) + tuple(local.shape[1:]))])
        g = _all_gather_rows(local)  # [N, n_pad, D, 3]: camera i N + r sits at g[r, i]
        ctx.cfg = (U, n_loc, n_pad, world)
        return g.transpose(0, 1).reshape((n_pad * world,) + tuple(local.shape[1:]))[:U].contiguous()

    @staticmethod
    def backward(ctx, d_cols):
        U, n_loc, n_pad, world = ctx.cfg
        tail = tuple(d_cols.shape[1:])
        if U < n_pad * world:
            d_cols = torch.cat([d_cols, d_cols.new_zeros((n_pad * world - U,) + tail)])
        full = d_cols.reshape((n_pad, world) + tail).transpose(0, 1).contiguous()  # [N, n_pad, ...]
        return _reduce_scatter_rows(full)[:n_loc], None, None, None
