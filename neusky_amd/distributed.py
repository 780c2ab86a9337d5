"""Ray-sharded data parallelism: one process per GPU, replicated parameters, ONE flat gradient
all-reduce (mean) per step over RCCL/xGMI (`torch.distributed` backend "nccl"; "gloo" in CPU tests).

Replaces `DDP(self._model, device_ids=[local_rank], find_unused_parameters=True)` + `dist.barrier`
at neusky/pipelines/neusky_pipeline.py:198-200.  `find_unused_parameters=True` semantics are kept by
zero-filling the slots of parameters that received no gradient (frozen RENI decoder, unused heads), so
ranks never disagree on the message.  The payload is dominated by the two 2^19 x 16 hash tables
(2 x 48.8 MB fp32); xGMI ring all-reduce of ~110 MB costs about a millisecond against a >15 ms step,
so a single bucket after backward is used rather than overlapped buckets.
"""
from __future__ import annotations

from typing import List

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
    """RCCL broadcasts device tensors in place; under gloo (two ranks sharing one GPU in tests) a device tensor goes through the host"""
    if t.is_cuda and dist.get_backend() != "nccl":
        host = t.cpu()
        dist.broadcast(host, src=src)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src)


class GradientAllReduce:
    """Per-parameter form of the gradient all-reduce (any list of parameters, gradients wherever autograd put them).  The
    training engine does not use it: `engine.Optimizers` keeps all gradients in one slab and all-reduces that in place
    (`Optimizers.all_reduce_gradients`).  The staging buffer here is therefore allocated on first use only."""

    def __init__(self, params: List[torch.nn.Parameter], world_size: int, module: torch.nn.Module = None):
        self.params = list(params)
        self.world_size = world_size
        self.module = module
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def broadcast_parameters(self, src: int = 0) -> None:
        if self.module is not None:
            broadcast_module_state(self.module, src)
            return
        for p in self.params:
            _broadcast(p.data, src)

    def barrier(self) -> None:
        dist.barrier()

    def all_reduce(self) -> None:
        if self.flat is None:
            self.flat = torch.zeros(self.numel, device=self.params[0].device, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(self.world_size)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                p.grad = self.flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(self.flat[off:off + n].view_as(p))
            off += n
