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
import torch.distributed as dist  # noqa: F401  (tests patch the exchange through this name)

from . import hip
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
        """flat_g: this group's range of the pipeline's gradient slab (distributed.GradientSlab: one zero fill, one all-reduce per
        step).  The parameters are re-homed into a slab of their own with the same layout, so an Adam step is one launch per group."""
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
            off += (k + 3) // 4 * 4
        if isinstance(self.sched, ExponentialDecaySchedulerConfig):
            self.sched.lr_init = self.opt.lr
        self.steps = 0


class Optimizers:
    """the five Adam groups as fused launches over the PIPELINE's gradient slab (neusky_amd.distributed.GradientSlab: layout, zero
    fill, collection and the all-reduce live there, behind `NeuSkyPipeline.get_train_loss_dict`; a trainer with torch optimizers
    uses the same slab through `p.grad`).  This class adds the parameter / moment slabs and the `nsky_adam_step` launches."""

    def __init__(self, config: Dict[str, Dict], param_groups: Dict[str, List[torch.nn.Parameter]], world_size: int = 1, slab=None):
        from .distributed import GradientSlab, slab_of
        self.groups = [_Group(k, param_groups[k], config[k]["optimizer"], config[k]["scheduler"])
                       for k in config if k in param_groups and len(param_groups[k]) > 0]
        self.world_size = world_size
        if slab is None:  # the slab the pipeline already built (world_size > 1, graph replay), else one over these groups
            slab = slab_of(self.groups[0].params)
        if slab is None:
            slab = GradientSlab({g.name: g.params for g in self.groups}, world_size, order=[g.name for g in self.groups])
        missing = [g.name for g in self.groups if g.name not in slab.group_views or slab.group_views[g.name].numel() != g.numel]
        if missing:
            raise ValueError(f"the gradient slab does not hold the optimizer groups {missing}")
        self.slab = slab
        self.flat_g = slab.flat
        for g in self.groups:
            g.bind(slab.group_views[g.name])
        slab.rebind()

    @property
    def _pending(self):
        return self.slab._pending

    def zero_grad_all(self) -> None:
        """zero the gradient slab.  Between this call and collect_grads() the .grad of a parameter that no kernel accumulates into the
        slab directly is None or an autograd-owned temporary: anything that reads .grad (clipping, logging, an optimizer step) must run
        AFTER collect_grads() -- all_reduce_gradients() and optimizer_scheduler_step_all() call it themselves."""
        self.slab.zero_all()

    def collect_grads(self) -> None:
        """after backward (inside the captured region of a graphed step): every parameter's .grad is its slab view again and
        holds the step's gradient"""
        self.slab.collect()

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

    def _buckets(self):
        """[(groups, slab view)] of the exchange (GradientSlab.buckets)"""
        by = {g.name: g for g in self.groups}
        return [([by[n] for n in names if n in by], view) for names, view in self.slab.buckets()]

    def all_reduce_gradients(self, bucketed: bool = False) -> None:
        """the gradient exchange (mean over ranks) -- GradientSlab.all_reduce -- unless the pipeline's end-of-pass hook has already run
        it for this pass (an eager `get_train_loss_dict` -> `backward` at world_size > 1 exchanges by itself, like DDP)."""
        self.collect_grads()  # (a no-op after train_iteration / a graphed step: a caller that ran backward() itself lands here)
        if self.slab.exchanged:
            return
        self.slab.all_reduce(bucketed)

    def _wait_bucket_of(self, group) -> None:
        self.slab.wait(group.name)  # nccl: the current stream waits for the collective (no host block); gloo: the host does

    def optimizer_scheduler_step_all(self, step: int) -> None:
        self.collect_grads()
        order = [g for gs, _ in self._buckets() for g in gs] if self.slab._pending else self.groups
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
    """One training iteration as a HIP-graph replay (the pipeline's own `TrainGraph`: zero-grad, forward, every loss and the whole
    backward are ONE graph launch) followed by the gradient all-reduce and the five fused Adam launches (outside the graph: the
    learning rates change every step).  A trainer with torch optimizers gets the same replay through
    `NeuSkyPipelineConfig.graph_replay` (pipelines/neusky_pipeline.py); this class is the fused-Adam client of it."""

    def __init__(self, pipeline, optimizers: Optimizers, ray_bundle, batch, warmup: int = 3, start_step: int = 0,
                 randoms: Optional[Dict] = None):
        from .pipelines.train_graph import TrainGraph
        self.pipeline, self.opt = pipeline, optimizers
        self.tg = TrainGraph(pipeline, optimizers.slab, ray_bundle, batch, warmup=warmup, start_step=start_step, randoms=randoms)

    graph = property(lambda self: self.tg.graph)
    loss = property(lambda self: self.tg.loss)
    loss_dict = property(lambda self: self.tg.loss_dict)
    metrics = property(lambda self: self.tg.metrics)
    randoms = property(lambda self: self.tg.randoms)
    rb = property(lambda self: self.tg.rb)
    batch = property(lambda self: self.tg.batch)
    sky = property(lambda self: self.tg.sky)

    def load(self, ray_bundle, batch, sky=None) -> None:
        self.tg.load(ray_bundle, batch, sky)

    def step(self, step: int, ray_bundle=None, batch=None, sky=None):
        self.tg.replay(step, ray_bundle, batch, sky)
        self.opt.all_reduce_gradients()
        self.opt.optimizer_scheduler_step_all(step)
        return self.tg.loss, self.tg.loss_dict, self.tg.metrics
