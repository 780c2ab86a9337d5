"""`SlabAdam`: torch.optim.Adam's update as ONE `nsky_adam_step` launch per optimizer, for a trainer that builds its optimizers from
configs (nerfstudio's `Optimizers`: one optimizer per parameter group, `optimizer.step()` / `zero_grad()` / `state_dict()`;
neusky/configs/neusky_config.py:216-237 -- five Adam groups, eps 1e-15).

What `neusky_amd.engine.Optimizers` does for the engine (parameters re-homed into one flat slab per group, moments in two more, one
launch per group) behind the `torch.optim.Optimizer` interface, so `ns-train neusky` gets the measured step whole: the graph replay
behind `NeuSkyPipeline.get_train_loss_dict` AND the fused Adam.  The update is torch.optim.Adam's (bias-corrected moments, eps added
to sqrt(v_hat), no weight decay, no amsgrad): `tests/test_gpu_slab_adam.py` compares the two step by step.

One difference, stated: torch's Adam skips a parameter whose `.grad` is None; here the whole slab is stepped with a zero gradient
for such a parameter.  With zero moments that is no movement -- the case of every parameter this model leaves unused in a step (they
are unused in every step); a parameter that trained and then stops receiving gradients would keep moving on its decaying first
moment, where torch's would freeze.
"""
from __future__ import annotations

from typing import Dict, List

import torch


class SlabAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, **unused):
        if weight_decay != 0.0:
            raise NotImplementedError("SlabAdam: weight decay is not part of the `neusky` optimizers (neusky_config.py:216-237)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0.0))
        self._slabs: List[Dict] = []
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.requires_grad]
            if not ps:
                self._slabs.append(None)
                continue
            dev = ps[0].device
            n = sum((p.numel() + 3) // 4 * 4 for p in ps)  # every parameter on a 16-byte boundary (the layout of distributed.GradientSlab)
            flat_p, m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
            flat_g = torch.zeros(n, device=dev)  # used when the gradients do not already sit in one slab range
            off, views = 0, []
            for p in ps:
                k = p.numel()
                flat_p[off:off + k].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + k].view_as(p)  # re-homed: the kernels of a captured step read the parameter at this address
                # per-parameter state in torch.optim.Adam's names (views of the slabs): state_dict() has Adam's shape
                self.state[p] = {"step": None, "exp_avg": m[off:off + k].view_as(p), "exp_avg_sq": v[off:off + k].view_as(p)}
                views.append((p, off, k))
                off += (k + 3) // 4 * 4
            step_t = torch.zeros((), dtype=torch.float32)  # ONE host counter per group, shared by its parameters' state entries
            for p in ps:
                self.state[p]["step"] = step_t
            self._slabs.append({"params": ps, "flat_p": flat_p, "m": m, "v": v, "flat_g": flat_g, "views": views, "steps": 0, "step_t": step_t,
                                "slab_range": None})
        from .distributed import slab_of
        slab = slab_of([p for g in self.param_groups for p in g["params"] if p.requires_grad])
        if slab is not None:
            slab.rebind()  # (the gradient sinks are keyed by the parameters' addresses, which have just changed)

    def _slab_range(self, st):
        """this group's range of the pipeline's GradientSlab when the slab lays the group out exactly like this optimizer does"""
        from .distributed import slab_of
        slab = slab_of(st["params"])
        if slab is None:
            return None
        by_id = {id(p): view for p, view in slab.views}
        first = by_id.get(id(st["params"][0]))
        if first is None:
            return None
        base = first.data_ptr()
        if not all(id(p) in by_id and by_id[id(p)].data_ptr() == base + 4 * off for p, off, _ in st["views"]):
            return None
        start = (base - slab.flat.data_ptr()) // 4
        n = st["flat_p"].numel()
        if start < 0 or start + n > slab.flat.numel():
            return None
        return slab.flat[start:start + n], by_id

    def _flat_gradient(self, st) -> torch.Tensor:
        """the group's gradient as one flat tensor in the slab's layout: the group's range of the pipeline's GradientSlab when every
        .grad is its view there or None (a parameter the step left unused: its slot is zero in the slab) -- no copy --, otherwise gathered
        into this optimizer's own buffer (one launch; missing gradients are zeros)"""
        from . import hip
        if st["slab_range"] is None:  # (probed until the pipeline has a slab: graph replay builds it at its first capture)
            st["slab_range"] = self._slab_range(st)
        if st["slab_range"] is not None:
            rng, by_id = st["slab_range"]
            if all(p.grad is None or p.grad.data_ptr() == by_id[id(p)].data_ptr() for p in st["params"]):
                return rng
        pairs, missing = [], False
        for p, off, k in st["views"]:
            g = p.grad
            if g is None:
                missing = True
                continue
            pairs.append((g.contiguous().reshape(-1), st["flat_g"][off:off + k]))
        if missing:
            st["flat_g"].zero_()
        hip.gather_segments(pairs)
        return st["flat_g"]

    @torch.no_grad()
    def step(self, closure=None):
        from . import hip
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, st in zip(self.param_groups, self._slabs):
            if st is None:
                continue
            g = self._flat_gradient(st)
            st["steps"] += 1
            b1, b2 = group["betas"]
            hip.adam_step(st["flat_p"], g, st["m"], st["v"], float(group["lr"]), float(b1), float(b2), float(group["eps"]), st["steps"])
            st["step_t"] += 1
        return loss

    def load_state_dict(self, state_dict) -> None:
        """torch's loader REPLACES the state tensors; here they are views of the moment slabs, so the loaded values are copied in"""
        keep = {id(p): (self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for st in self._slabs if st for p in st["params"]}
        super().load_state_dict(state_dict)
        for st in self._slabs:
            if st is None:
                continue
            steps = 0
            for p in st["params"]:
                s = self.state.get(p)
                m_view, v_view = keep[id(p)]
                if s is not None and "exp_avg" in s:
                    m_view.copy_(s["exp_avg"])
                    v_view.copy_(s["exp_avg_sq"])
                    steps = max(steps, int(s["step"]))
            st["steps"] = steps
            st["step_t"].fill_(float(steps))
            for p in st["params"]:
                m_view, v_view = keep[id(p)]
                self.state[p] = {"step": st["step_t"], "exp_avg": m_view, "exp_avg_sq": v_view}
