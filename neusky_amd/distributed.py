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
    """What is left of the DDP wrapper once the gradients live in `engine.Optimizers`' slab (which all-reduces itself,
    `Optimizers.all_reduce_gradients`): the initial state broadcast and the barrier of neusky_pipeline.py:198-200."""

    def __init__(self, module: torch.nn.Module, world_size: int):
        self.module = module
        self.world_size = world_size

    def broadcast_parameters(self, src: int = 0) -> int:
        return broadcast_module_state(self.module, src)

    def barrier(self) -> None:
        dist.barrier()
