"""The train step of `NeuSkyPipeline.get_train_loss_dict` (neusky/pipelines/neusky_pipeline.py:241-291) plus the backward pass the trainer
runs on its result, captured once as a HIP graph and replayed: zero-grad, forward, every loss and the whole backward are ONE graph
launch, so the ~1100 kernel launches of a step cost no host time.  What stays outside is what changes per step or belongs to the
trainer: loading the next batch into the graph's static buffers (one launch), the gradient exchange, the optimizer.

Everything inside the graph has static shapes and no host dependency: the step's inputs live in fixed device buffers (`load`), the
illumination-direction rotation, sample jitter, hash-grid probe, vMF DDF rays and multi-view points are drawn on the device, every
training camera's illumination is decoded (no torch.unique), and the upper-hemisphere direction subset has the static size D/2
(antipodal direction set).

Two clients: `NeuSkyPipeline` itself (config `graph_replay`: any trainer that calls get_train_loss_dict -> sum -> backward ->
optimizer.step gets the replay; the returned loss carries an autograd node whose backward hands the slab views to `p.grad`), and
`neusky_amd.engine.GraphedTrainStep` (replay + fused Adam).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import hip, ops
from ..model_components.losses import LossDict, total_loss


def _detached(v):
    if torch.is_tensor(v):
        return v.detach()
    if isinstance(v, dict):
        return {k: _detached(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_detached(x) for x in v)
    return v


class _ReattachFn(torch.autograd.Function):
    """the autograd seam of a replayed step: forward aliases a static loss scalar of the graph, backward -- the trainer's
    `loss.backward()` -- makes `p.grad` the slab views the replay has already filled (TrainGraph.reattach)"""

    @staticmethod
    def forward(ctx, anchor, value, graph):
        ctx.graph = graph
        return value.view_as(value)

    @staticmethod
    def backward(ctx, g):
        ctx.graph.reattach(g)
        return None, None, None


class TrainGraph:
    def __init__(self, pipeline, slab, ray_bundle, batch, warmup: int = 1, start_step: int = 0, randoms: Optional[Dict] = None):
        """randoms: optional injected random draws (tests): cloned into static device buffers the graph reads on every
        replay, so a replay can be compared with the eager step / the oracle on the same draws."""
        from ..cameras.rays import RayBundle
        self.pipeline, self.slab = pipeline, slab
        dev = ray_bundle.origins.device
        c = lambda t: t.detach().clone()  # noqa: E731
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
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self._attached = True
        side = ops.role_stream("capture", dev)  # warm-up and capture on the package's capture stream (ops.role_stream: never an alias of a role stream)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # eager warm-up on the side stream: caches, autotuned paths, allocator pools
            for i in range(warmup):
                self._body(start_step + i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other host threads (e.g. the RCCL watchdog) may legally touch the runtime during the capture
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode=ops.CAPTURE_MODE):
            self.outputs, self.loss, self.loss_dict, self.metrics = self._body(start_step + warmup)
        self.used = list(slab.used)  # which parameters the captured pass gives a gradient
        torch.cuda.synchronize()

    def __del__(self):
        # a pipeline dropped right behind its last replay: the graph is retired, not destroyed here (ops.retire_graph)
        try:
            ops.retire_graph(self.__dict__.pop("graph", None))
        except Exception:  # noqa: BLE001  (interpreter shutdown: modules may be gone; the process is ending anyway)
            pass

    def _body(self, step):
        self.slab.zero_all()
        outputs, loss_dict, metrics = self.pipeline._train_loss_dict(step, ray_bundle=self.rb, batch=self.batch, randoms=self.randoms)
        loss = total_loss(loss_dict)
        loss.backward()
        self.slab.collect()
        return _detached(outputs), loss.detach(), {k: v.detach() for k, v in loss_dict.items()}, _detached(metrics)

    def load(self, ray_bundle, batch, sky=None) -> None:
        """the next step's inputs into the graph's static buffers: ONE launch for all of them when they are already on the device
        (hip.copy_segments), otherwise a copy per tensor (host batches)"""
        pairs = [(ray_bundle.origins, self.rb.origins), (ray_bundle.directions, self.rb.directions),
                 (ray_bundle.camera_indices, self.rb.camera_indices)]
        pairs += [(v, self.rb.metadata[k]) for k, v in ray_bundle.metadata.items() if k in self.rb.metadata]
        pairs += [(batch["image"], self.batch["image"]), (batch["mask"], self.batch["mask"])]
        if sky is not None:
            pairs += [(sky.origins, self.sky.origins), (sky.directions, self.sky.directions)]
        if all(s.device == d.device and s.is_contiguous() and d.is_contiguous() and s.dtype == d.dtype and s.shape == d.shape for s, d in pairs):
            hip.copy_segments(pairs)
        else:
            for s, d in pairs:
                d.copy_(s.reshape(d.shape), non_blocking=True)

    def load_randoms(self, randoms: Dict) -> None:
        """injected random draws of the next step into the static buffers the capture was given (same keys and shapes)"""
        def put(dst, src):
            if torch.is_tensor(dst):
                dst.copy_(src.to(dst.device).reshape(dst.shape), non_blocking=True)
            elif isinstance(dst, (list, tuple)):
                for d, s_ in zip(dst, src):
                    put(d, s_)
            elif hasattr(dst, "origins") and hasattr(src, "origins"):  # a RayBundle (the DDF-fit rays)
                put(dst.origins, src.origins)
                put(dst.directions, src.directions)
        for k, v in randoms.items():
            if k == "sky_ray_bundle":
                put(self.sky, v)
            elif k in self.randoms:
                put(self.randoms[k], v)
            else:
                raise KeyError(f"the captured step was not given a static buffer for the random draw {k!r}")

    def replay(self, step: int, ray_bundle=None, batch=None, sky=None, randoms: Optional[Dict] = None) -> None:
        if ray_bundle is not None:
            self.load(ray_bundle, batch, sky)
        if randoms is not None:
            self.load_randoms(randoms)
        self.pipeline.model.set_step(step)  # proposal-weight anneal: a device scalar the graph reads
        self.graph.replay()
        self.slab.exchanged = False
        self._attached = False

    # ------------------------------------------------------------------ trainer form
    def trainer_loss_dict(self) -> LossDict:
        """the loss dict a trainer sums and backpropagates: the graph's static scalars; the first entry and `.total` each carry the
        re-attach node (a trainer adds the entries up -- nerfstudio's -- or asks total_loss())"""
        out = LossDict(self.loss_dict)
        for k in out:
            out[k] = _ReattachFn.apply(self._anchor, out[k], self)
            break
        out.total = _ReattachFn.apply(self._anchor, self.loss, self)
        return out

    def reattach(self, grad_output=None) -> None:
        """`p.grad` <- the slab views of the parameters the captured pass gives a gradient (the trainer's zero_grad dropped them);
        with an enabled GradScaler the slab takes the loss scale the trainer backpropagates"""
        if self._attached:
            return
        self._attached = True
        scaler = getattr(self.pipeline, "grad_scaler", None)
        if grad_output is not None and scaler is not None and scaler.is_enabled():
            self.slab.flat.mul_(grad_output.reshape(()))
        for (p, view), used in zip(self.slab.views, self.used):
            p.grad = view if used else None
