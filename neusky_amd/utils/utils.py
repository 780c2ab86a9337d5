"""Small geometry / colour helpers (torch ops, small tensors): neusky/utils/utils.py."""
from __future__ import annotations

import torch


def linear_to_sRGB(color: torch.Tensor, use_quantile: bool = False) -> torch.Tensor:
    """neusky/utils/utils.py:11-31"""
    if use_quantile:
        q = torch.quantile(color.flatten(), 0.98)
        color = color / q.expand_as(color)
    color = torch.where(color <= 0.0031308, 12.92 * color, 1.055 * torch.pow(torch.abs(color), 1 / 2.4) - 0.055)
    return torch.clamp(color, 0.0, 1.0)


def ray_sphere_intersection(positions: torch.Tensor, directions: torch.Tensor, radius: float) -> torch.Tensor:
    """neusky/utils/utils.py:68-93 (unit directions, far root)"""
    b = 2 * (directions * positions).sum(-1)
    c = (positions * positions).sum(-1) - radius**2
    disc = b**2 - 4 * c
    t = torch.max((-b - torch.sqrt(disc)) / 2, (-b + torch.sqrt(disc)) / 2)
    return positions + t[..., None] * directions


def sph2cart(theta, phi):
    """neusky/utils/utils.py:95-99"""
    return torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)


def rot_z(gamma: torch.Tensor) -> torch.Tensor:
    """neusky/utils/utils.py:168-173"""
    c, s = torch.cos(gamma), torch.sin(gamma)
    return torch.tensor([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=gamma.dtype)


def to_device_async(t: torch.Tensor, device) -> torch.Tensor:
    """host tensor -> device through pinned memory with a non-blocking copy (a pageable .to(device) blocks the host
    until the stream drains, which serialises the CPU-side generators of a step with the GPU work)"""
    if t.device.type != "cpu" or str(device) == "cpu":
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def device_rng_seed(stream_id: int = 0, base: int = None) -> int:
    """seed of an in-kernel counter-based generator (csrc/samplers.hip): torch's seed (or `base`, a seed restored from a checkpoint),
    the rank (ranks that were seeded alike must not draw the same rays) and a per-generator stream id"""
    b = torch.initial_seed() + 104729 * stream_id if base is None else int(base)
    return (b + 7919 * _rank()) & (2**63 - 1)


# ---- in-kernel generators: (seed, call counter on the device) held by the object that draws, listed here for checkpoints
import weakref

_RNG_OWNERS: dict = {}    # name -> weak reference to the object holding the generator's state
_RNG_PENDING: dict = {}   # name -> (seed, calls) loaded from a checkpoint before the generator exists


def device_rng(owner, name: str, stream_id: int, device) -> tuple:
    """(seed, int64 device counter) of `owner`'s in-kernel generator `name`: created at its first use (seed: device_rng_seed, counter 0)
    or, after utils.checkpoints.load_checkpoint, resumed from the saved seed and call count -- a resumed run continues the random
    sequence instead of redrawing step 0's."""
    attr = "_nsky_rng_" + name
    st = getattr(owner, attr, None)
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:  # "cuda" and "cuda:0" name the same device: never re-create the state (and its counter)
        dev = torch.device("cuda", torch.cuda.current_device())
    if st is None or st[1].device != dev:
        pend = _RNG_PENDING.pop(name, None)
        # a checkpoint holds the rank-free base of the seed: every rank re-derives its own (ranks never share a random sequence)
        st = (device_rng_seed(stream_id) if pend is None else device_rng_seed(stream_id, base=pend[0]), torch.zeros(1, dtype=torch.int64, device=dev))
        if pend is not None:
            st[1].fill_(int(pend[1]))
        object.__setattr__(owner, attr, st)
        _RNG_OWNERS[name] = weakref.ref(owner)
    return st


def device_rng_state() -> dict:
    out = {}
    for name, ref in _RNG_OWNERS.items():
        o = ref()
        st = getattr(o, "_nsky_rng_" + name, None) if o is not None else None
        if st is not None:  # (rank-free base of the seed, calls so far)
            out[name] = ((int(st[0]) - 7919 * _rank()) & (2**63 - 1), int(st[1].item()))
    return out


def load_device_rng_state(state: dict) -> None:
    for name, (base, count) in state.items():
        ref = _RNG_OWNERS.get(name)
        o = ref() if ref is not None else None
        st = getattr(o, "_nsky_rng_" + name, None) if o is not None else None
        if st is not None:  # a live generator: same counter tensor (a captured graph reads it); the checkpoint's seed base + THIS rank
            object.__setattr__(o, "_nsky_rng_" + name, (device_rng_seed(0, base=base), st[1].fill_(int(count))))
        else:
            _RNG_PENDING[name] = (int(base), int(count))
