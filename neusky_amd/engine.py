"""Optimizer groups + one train iteration.

Replaces, for the hot path only, what nerfstudio's `Trainer.train_iteration` / `Optimizers` do around
`pipeline.get_train_loss_dict`: sum the loss dict, backward, step the five Adam
groups of neusky/configs/neusky_config.py:216-237 with their schedulers.  Each group's parameters and
gradients live in ONE flat fp32 slab (parameters are re-homed as views), so a step is one
`nsky_adam_step` launch per group and the multi-GPU gradient all-reduce runs on the slabs directly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import hip, ops
from .model_components.losses import total_loss



@dataclass
class AdamOptimizerConfig:
    lr: float = 1e-3
    eps: float = 1e-15
    betas: tuple = (0.9, 0.999)


@dataclass
class CosineDecaySchedulerConfig:
    warm_up_end: int = 500
    learning_rate_alpha: float = 0.05
    max_steps: int = 100001

    def factor(self, step: int) -> float:
        if step < self.warm_up_end:
            return step / self.warm_up_end
        progress = (step - self.warm_up_end) / (self.max_steps - self.warm_up_end)
        return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - self.learning_rate_alpha) + self.learning_rate_alpha


@dataclass
class ExponentialDecaySchedulerConfig:
    lr_final: Optional[float] = None
    max_steps: int = 100001
    warmup_steps: int = 0
    lr_pre_warmup: float = 1e-8
    lr_init: float = 1.0  # filled from the optimizer

    def factor(self, step: int) -> float:
        lr_init = self.lr_init
        lr_final = lr_init if self.lr_final is None else self.lr_final
        if step < self.warmup_steps:
            lr = self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * math.sin(0.5 * math.pi * min(max(step / self.warmup_steps, 0), 1))
        else:
            t = min(max((step - self.warmup_steps) / (self.max_steps - self.warmup_steps), 0), 1)
            lr = math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
        return lr / lr_init


def neusky_optimizers() -> Dict[str, Dict]:
    """neusky/configs/neusky_config.py:216-237"""
    return {
        "proposal_networks": {"optimizer": AdamOptimizerConfig(lr=1e-2), "scheduler": CosineDecaySchedulerConfig()},
        "fields": {"optimizer": AdamOptimizerConfig(lr=1e-3), "scheduler": CosineDecaySchedulerConfig()},
        "illumination_field": {"optimizer": AdamOptimizerConfig(lr=1e-2), "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-5)},
        "visibility_sigmoid": {"optimizer": AdamOptimizerConfig(lr=1e-3),
                               "scheduler": ExponentialDecaySchedulerConfig(warmup_steps=4000, lr_final=1e-4)},
        "ddf_field": {"optimizer": AdamOptimizerConfig(lr=1e-4), "scheduler": CosineDecaySchedulerConfig()},
    }


class _Group:
    def __init__(self, name, params: List[torch.nn.Parameter], opt: AdamOptimizerConfig, sched):
        self.name, self.opt, self.sched = name, opt, sched
        self.params = [p for p in params if p.requires_grad]
        # every parameter starts on a 16-byte boundary of the slab (the GEMM operands are read with float4 loads)
        self.numel = sum((p.numel() + 3) // 4 * 4 for p in self.params)

    def bind(self, flat_g: torch.Tensor) -> None:
        """flat_g: this group's slice of the ONE gradient slab all groups share (one zero fill, one all-reduce per step)"""
        n = self.numel
        dev = self.params[0].device
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = flat_g
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            p.grad = self.flat_g[off:off + k].view_as(p)
            p._nsky_grad_sink = True  # custom backward passes may accumulate into p.grad directly (zeroed by zero_grad_all)
            ops.register_grad_sink(p)
            off += (k + 3) // 4 * 4
        if isinstance(self.sched, ExponentialDecaySchedulerConfig):
            self.sched.lr_init = self.opt.lr
        self.steps = 0


class Optimizers:
    def __init__(self, config: Dict[str, Dict], param_groups: Dict[str, List[torch.nn.Parameter]], world_size: int = 1):
        self.groups = [_Group(k, param_groups[k], config[k]["optimizer"], config[k]["scheduler"])
                       for k in config if k in param_groups and len(param_groups[k]) > 0]
        self.world_size = world_size
        # ONE gradient slab for all groups: a step zero-fills it once and all-reduces it once (~110 MB, dominated by the two
        # hash tables) instead of five messages of which three are latency-only (the visibility threshold is one float)
        self.flat_g = torch.zeros(sum(g.numel for g in self.groups), device=self.groups[0].params[0].device)
        off = 0
        for g in self.groups:
            g.bind(self.flat_g[off:off + g.numel])
            off += g.numel
        self._views = [(p, p.grad) for g in self.groups for p in g.params]
        self._backward_seen = False
        # The ONE host seam of this class: a slab in host memory exists only in the world-size-2 `gloo` tests of the exchange logic
        # (tests/test_cpu_distributed.py: layout, buckets, zero-fill semantics, mean) -- there gradients are accumulated by autograd
        # into the slab views and nothing is gathered or stepped; every training path has the slab in HBM.
        self._host_slab = self.flat_g.device.type == "cpu"

    def zero_grad_all(self) -> None:
        """zero the gradient slab.  Between this call and collect_grads() the .grad of a parameter that no kernel accumulates into the
        slab directly is None or an autograd-owned temporary: anything that reads .grad (clipping, logging, an optimizer step) must run
        AFTER collect_grads() -- all_reduce_gradients() and optimizer_scheduler_step_all() call it themselves."""
        self.flat_g.zero_()
        # Parameters whose gradient no kernel writes into the slab itself (everything behind weight norm / padding / plain torch
        # ops) start the backward with an undefined .grad: autograd's AccumulateGrad then keeps the incoming tensor instead of
        # launching one add kernel per parameter, and collect_grads() moves all of them into the slab with one launch.
        if self._backward_seen and not self._host_slab:
            for p, _ in self._views:
                if not getattr(p, "_nsky_sunk", False):
                    p.grad = None

    def collect_grads(self) -> None:
        """after backward (inside the captured region of a graphed step): every parameter's .grad is its slab view again and
        holds the step's gradient"""
        self._backward_seen = True
        pairs = []
        for p, view in self._views:
            g = p.grad
            if g is not None and g.data_ptr() != view.data_ptr():
                pairs.append((g if g.is_contiguous() else g.contiguous(), view))
            p.grad = view
        if pairs:
            hip.gather_segments(pairs)

    def state_dict(self) -> Dict[str, Dict]:
        """per group: Adam moments and the bias-correction step count (what nerfstudio's trainer keeps under "optimizers" /
        "schedulers" in step-%09d.ckpt; the schedulers here are pure functions of the global step).  The parameters themselves
        travel in the pipeline's state dict: on load they are copied IN PLACE into the flat slabs they are views of."""
        return {g.name: {"m": g.m.detach().clone(), "v": g.v.detach().clone(), "steps": g.steps, "numel": g.numel}
                for g in self.groups}

    def load_state_dict(self, state: Dict[str, Dict]) -> None:
        for g in self.groups:
            if g.name not in state:
                raise KeyError(f"optimizer state has no group {g.name!r} (has {sorted(state)})")
            st = state[g.name]
            if int(st["numel"]) != g.numel:
                raise ValueError(f"optimizer group {g.name!r}: {st['numel']} saved elements, {g.numel} expected")
            g.m.copy_(st["m"].to(g.m.device))
            g.v.copy_(st["v"].to(g.v.device))
            g.steps = int(st["steps"])

    # buckets of the gradient exchange, in the order the backward pass finishes them: the DDF group (its chain backward and hash-table
    # scatter run first) with the scalar / latent groups, then the field and proposal groups (last to finish).  Segments of one bucket are
    # adjacent in the slab when the groups are in this order; otherwise the bucket falls back to per-group messages.
    COMM_BUCKETS = (("ddf_field", "visibility_sigmoid", "illumination_field"), ("fields", "proposal_networks"))

    def _buckets(self):
        """[(groups, slab view)]: the contiguous slab range covered by the groups of each bucket that exist"""
        out = []
        offs, off = {}, 0
        for g in self.groups:
            offs[g.name] = (off, off + g.numel)
            off += g.numel
        done = set()
        for names in self.COMM_BUCKETS:
            gs = [g for g in self.groups if g.name in names]
            spans = sorted(offs[g.name] for g in gs)
            if gs and all(a[1] == b[0] for a, b in zip(spans, spans[1:])):
                out.append((gs, self.flat_g[spans[0][0]:spans[-1][1]]))
            else:
                out += [([g], g.flat_g) for g in gs]
            done |= {g.name for g in gs}
        out += [([g], g.flat_g) for g in self.groups if g.name not in done]
        return out

    def all_reduce_gradients(self, bucketed: bool = False) -> None:
        """the gradient exchange (mean over ranks) over RCCL / xGMI: ONE all-reduce of the whole slab (default: ~110 MB, 0.5-1.3 ms
        on 8 x MI355X against a 20 ms step), or -- bucketed -- one asynchronous all-reduce per bucket, issued back
        to back on RCCL's own stream; optimizer_scheduler_step_all then makes each group's Adam launch wait for its own bucket only,
        so the second bucket's exchange overlaps the first bucket's Adam steps."""
        self.collect_grads()  # (a no-op after train_iteration / a graphed step: a caller that ran backward() itself lands here)
        self._pending = []
        if self.world_size <= 1:
            return
        backend = dist.get_backend()
        if backend == "nccl":
            if bucketed:
                self._pending = [(gs, dist.all_reduce(view, op=dist.ReduceOp.AVG, async_op=True)) for gs, view in self._buckets()]
            else:
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.AVG)
        elif not self._host_slab:  # gloo with device gradients (two ranks sharing one GPU in tests): staged through the host
            host = self.flat_g.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            self.flat_g.copy_(host.div_(self.world_size))
        elif bucketed:  # gloo (CPU tests) has no AVG: asynchronous SUMs, the division happens when the bucket is waited for
            self._pending = [(gs, dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view) for gs, view in self._buckets()]
        else:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM)
            self.flat_g.div_(self.world_size)

    def _wait_bucket_of(self, group) -> None:
        for item in list(getattr(self, "_pending", [])):
            if any(g is group for g in item[0]):
                item[1].wait()  # nccl: the current stream waits for the collective (no host block); gloo: the host does
                if len(item) == 3:
                    item[2].div_(self.world_size)
                self._pending.remove(item)

    def optimizer_scheduler_step_all(self, step: int) -> None:
        self.collect_grads()
        order = [g for gs, _ in self._buckets() for g in gs] if getattr(self, "_pending", None) else self.groups
        for g in order:
            self._wait_bucket_of(g)
            g.steps += 1
            lr = g.opt.lr * (g.sched.factor(step) if g.sched is not None else 1.0)
            hip.adam_step(g.flat_p, g.flat_g, g.m, g.v, lr, g.opt.betas[0], g.opt.betas[1], g.opt.eps, g.steps)


def train_iteration(pipeline, optimizers: Optimizers, step: int, **kw):
    """one step of the reference's training loop: forward + losses, backward, (all-reduce), 5 Adam steps"""
    optimizers.zero_grad_all()
    _, loss_dict, metrics_dict = pipeline.get_train_loss_dict(step, **kw)
    loss = total_loss(loss_dict)
    loss.backward()
    optimizers.collect_grads()
    optimizers.all_reduce_gradients()
    optimizers.optimizer_scheduler_step_all(step)
    return loss.detach(), loss_dict, metrics_dict


class GraphedTrainStep:
    """One training iteration captured in a HIP graph (torch.cuda.graph): zero-grad, forward, every loss and the whole
    backward replay as ONE graph launch, so the ~1100 kernel launches of a step cost no host time; the gradient
    all-reduce and the five Adam launches stay outside (the learning rates change every step).

    Everything inside the graph has static shapes and no host dependency: the step's inputs live in fixed device
    buffers (`load`), the illumination-direction rotation, sample jitter, hash-grid probe, vMF DDF rays and multi-view
    points are drawn on the device, every training camera's illumination is decoded (no torch.unique), and the
    upper-hemisphere direction subset has the static size D/2 (antipodal direction set)."""

    def __init__(self, pipeline, optimizers: Optimizers, ray_bundle, batch, warmup: int = 3, start_step: int = 0,
                 randoms: Optional[Dict] = None):
        """randoms: optional injected random draws (tests): cloned into static device buffers the graph reads on every
        replay, so a replay can be compared with the eager step / the oracle on the same draws."""
        from .cameras.rays import RayBundle
        self.pipeline, self.opt = pipeline, optimizers
        dev = ray_bundle.origins.device
        c = lambda t: t.detach().clone()
        self.rb = RayBundle(origins=c(ray_bundle.origins), directions=c(ray_bundle.directions), pixel_area=c(ray_bundle.pixel_area),
                            camera_indices=c(ray_bundle.camera_indices), metadata={k: c(v) for k, v in ray_bundle.metadata.items()})
        self.batch = {"image": c(batch["image"]), "mask": c(batch["mask"])}
        sky = pipeline.datamanager.get_sky_ray_bundle(pipeline.config.num_sky_rays)
        self.sky = RayBundle(origins=c(sky.origins), directions=c(sky.directions))
        self.randoms = {"sky_ray_bundle": self.sky}
        if randoms is not None:
            def static(v):
                if torch.is_tensor(v):
                    return c(v.to(dev))
                if isinstance(v, (list, tuple)):
                    return type(v)(static(x) for x in v)
                return v
            self.randoms.update({k: static(v) for k, v in randoms.items() if k != "sky_ray_bundle"})
        self.step_idx = start_step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager warm-up on the side stream: caches, autotuned paths, allocator pools
            for i in range(warmup):
                self._body(start_step + i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other host threads (e.g. the RCCL watchdog) may legally touch the runtime during the capture
        with torch.cuda.graph(self.graph, capture_error_mode=ops.CAPTURE_MODE):
            self.loss, self.loss_dict, self.metrics = self._body(start_step + warmup)
        torch.cuda.synchronize()

    def _body(self, step):
        self.opt.zero_grad_all()
        _, loss_dict, metrics = self.pipeline.get_train_loss_dict(step, ray_bundle=self.rb, batch=self.batch, randoms=self.randoms)
        loss = total_loss(loss_dict)
        loss.backward()
        self.opt.collect_grads()
        return loss.detach(), {k: v.detach() for k, v in loss_dict.items()}, metrics

    def load(self, ray_bundle, batch, sky=None) -> None:
        """the next step's inputs into the graph's static buffers: ONE launch for all of them when they are already on the device
        (hip.copy_segments), otherwise a copy per tensor (host batches)"""
        pairs = [(ray_bundle.origins, self.rb.origins), (ray_bundle.directions, self.rb.directions),
                 (ray_bundle.camera_indices, self.rb.camera_indices)]
        pairs += [(v, self.rb.metadata[k]) for k, v in ray_bundle.metadata.items()]
        pairs += [(batch["image"], self.batch["image"]), (batch["mask"], self.batch["mask"])]
        if sky is not None:
            pairs += [(sky.origins, self.sky.origins), (sky.directions, self.sky.directions)]
        if all(s.device == d.device and s.is_contiguous() and d.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape for s, d in pairs):
            hip.copy_segments(pairs)
        else:
            for s, d in pairs:
                d.copy_(s, non_blocking=True)

    def step(self, step: int, ray_bundle=None, batch=None, sky=None):
        if ray_bundle is not None:
            self.load(ray_bundle, batch, sky)
        self.pipeline.model.set_step(step)  # proposal-weight anneal: a device scalar the graph reads
        self.graph.replay()
        self.opt.all_reduce_gradients()
        self.opt.optimizer_scheduler_step_all(step)
        return self.loss, self.loss_dict, self.metrics
