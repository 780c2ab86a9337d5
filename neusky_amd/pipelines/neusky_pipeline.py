"""Train-step orchestration - mirrors `neusky.pipelines.neusky_pipeline.NeuSkyPipeline`
(neusky/pipelines/neusky_pipeline.py:99-515) for `__init__` (:117-202), `get_param_groups` (:227-238),
`get_train_loss_dict` (:241-291), `generate_ddf_samples` (:493-515), `_setup_visibility_field` (:446-491).

Multi-GPU: one process per GPU; parameters are replicated and the gradients are all-reduced (mean) over RCCL once per step, BY THE
PIPELINE: at world_size > 1 every gradient lives in one flat slab (`neusky_amd.distributed.GradientSlab`, `p.grad` are views of it)
and the loss dict `get_train_loss_dict` returns carries an autograd node whose backward queues an end-of-pass callback that collects
the gradients into the slab and all-reduces it -- what DDP's reducer does for the reference (:198-199).  So nerfstudio's trainer loop
(`zero_grad_all -> get_train_loss_dict -> sum -> backward -> optimizer.step`) sees synchronised `p.grad` with torch optimizers, and
`neusky_amd.engine` (fused Adam over the same slab) is one more client.  The reference wraps the model in DDP and then dereferences
`.visibility_field` on the wrapper (:249-250, :278), which does not work; here the model is never wrapped, so the same
call sites work at any world size.

`graph_replay` (config, off by default like every non-reference key): after `graph_replay_warmup` eager calls, `get_train_loss_dict`
captures forward + losses + backward in a HIP graph (pipelines/train_graph.py) and from then on loads the datamanager's next batch
into the graph's static buffers, replays, exchanges, and returns a loss whose `.backward()` only hands the slab views to `p.grad`.
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Type

import torch
from torch import nn
from torch.nn import Parameter

from ..data.synthetic_datamanager import SyntheticDataManagerConfig
from ..model_components.ddf_sampler import VMFDDFSamplerConfig
from ..model_components.losses import merge_loss_dicts
from ..models.ddf_model import DDFModelConfig
from ..models.neusky_model import NeuSkyFactoModelConfig
from ..plugin import ConfigBase, PipelineBase


@dataclass
class NeuSkyPipelineConfig(ConfigBase):
    """neusky/pipelines/neusky_pipeline.py:61-96 with the `neusky` values (neusky_config.py:43-215)"""

    _target: Type = field(default_factory=lambda: NeuSkyPipeline)
    datamanager: Any = field(default_factory=SyntheticDataManagerConfig)
    model: NeuSkyFactoModelConfig = field(default_factory=NeuSkyFactoModelConfig)
    visibility_field: Optional[DDFModelConfig] = field(default_factory=DDFModelConfig)
    visibility_field_radius: Any = "AABB"
    visibility_train_sampler: VMFDDFSamplerConfig = field(default_factory=VMFDDFSamplerConfig)
    visibility_accumulation_mask_threshold: float = 0.0
    num_sky_rays: int = 256
    test_mode: Optional[str] = None
    stop_sdf_gradients: bool = False
    least_squares_global_scale: bool = False
    graph_replay: bool = False  # (not a reference key) the train step as a HIP-graph replay behind get_train_loss_dict
    graph_replay_warmup: int = 3  # eager calls before the capture
    bucketed_exchange: bool = False  # (not a reference key) the all-reduce as two asynchronous buckets instead of one message
    shard_illumination_decode: bool = False  # (not a reference key) world_size > 1: rank r decodes cameras r::N (distributed.CameraAllGather)

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class _ExchangeFn(torch.autograd.Function):
    """identity; its backward -- the first node of the trainer's backward pass -- queues the pipeline's end-of-pass exchange"""

    @staticmethod
    def forward(ctx, x, pipe_ref):
        ctx.pipe_ref = pipe_ref
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        pipe = ctx.pipe_ref()
        if pipe is not None:
            pipe._queue_exchange()
        return g, None


class NeuSkyPipeline(PipelineBase):
    def __init__(self, config: NeuSkyPipelineConfig, device: str, test_mode: str = "val", world_size: int = 1,
                 local_rank: int = 0, grad_scaler=None):
        nn.Module.__init__(self)  # not the nerfstudio base's constructor (neusky_amd/plugin.py)
        self.config = config
        self.test_mode = test_mode if config.test_mode is None else config.test_mode
        self.datamanager = config.datamanager.setup(device=device, test_mode=self.test_mode, world_size=world_size, local_rank=local_rank,
                                                    eval_latent_optimise_method=config.model.eval_latent_optimise_method)  # :129-135
        assert self.datamanager.train_dataset is not None, "Missing input dataset"
        self.register_buffer("num_train_data", torch.tensor(len(self.datamanager.train_dataset)))  # :146-148
        self.register_buffer("num_test_data", torch.tensor(self.datamanager.num_test))
        self.register_buffer("num_val_data", torch.tensor(self.datamanager.num_val))
        self.scene_box = self.datamanager.train_dataset.scene_box
        visibility_field = None
        if config.visibility_field is not None:
            visibility_field = self._setup_visibility_field(device=device)
            self.visibility_train_sampler = config.visibility_train_sampler.setup(
                ddf_sphere_radius=visibility_field.ddf_radius, device=device)
        self._model = config.model.setup(
            scene_box=self.scene_box, num_train_data=int(self.num_train_data), num_val_data=int(self.num_val_data),
            num_test_data=int(self.num_test_data), visibility_field=visibility_field, test_mode=test_mode,
            train_metadata=self.datamanager.train_dataset.metadata, eval_metadata=self.datamanager.eval_dataset.metadata,
            grad_scaler=grad_scaler)
        self._model.to(device)
        self.world_size, self.local_rank = world_size, local_rank
        self.step_of_last_latent_optimisation = 0  # neusky_pipeline.py:202
        self.eval_image_num = 0
        self.max_eval_num = max(int(self.num_val_data if self.test_mode == "val" else self.num_test_data), 1)
        self.grad_scaler = grad_scaler
        self.grad_sync = None
        self._slab = None          # distributed.GradientSlab: built here at world_size > 1, else on first use (graph replay, engine)
        self._train_graph = None   # train_graph.TrainGraph once captured (config.graph_replay)
        self._train_calls = 0
        self._exchange_queued = False
        # True at world_size > 1.  (A one-rank process group -- the RCCL self-check of bench.py and tests -- may set it to run the
        # exchange with one rank: an AVG all-reduce over one rank is the identity.)
        self.exchange_gradients = world_size > 1
        if world_size > 1:
            from ..distributed import ReplicaSync
            self.grad_sync = ReplicaSync(self, world_size)
            self.grad_sync.broadcast_parameters()  # identical replicas (all parameters, frozen ones too, and buffers), then the :200 barrier
            self.gradient_slab()  # the reducer of :198-199
            if config.shard_illumination_decode:
                import torch.distributed as dist
                self._model.illumination_shard = (dist.get_rank(), world_size)
                self._model.illumination_sampler.shared_across_ranks = True  # one direction set per step on all ranks
            self.grad_sync.barrier()

    @property
    def model(self):
        return self._model

    def _setup_visibility_field(self, device):
        """:446-491 (checkpoint branch = SURVEY 8(f) item 3)"""
        aabb = self.scene_box["aabb"] if isinstance(self.scene_box, dict) else self.scene_box.aabb
        radius = float(torch.abs(aabb[0, 0])) if self.config.visibility_field_radius == "AABB" else float(self.config.visibility_field_radius)
        vf = self.config.visibility_field.setup(scene_box=None, num_train_data=int(self.num_train_data), ddf_radius=radius)
        vf.to(device)
        vf.train(self.config.model.fit_visibility_field)
        return vf

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        """:227-238 -> keys fields, proposal_networks, illumination_field, visibility_sigmoid, ddf_field"""
        groups = {**self.datamanager.get_param_groups(), **self.model.get_param_groups()}
        if self.model.visibility_field is not None:
            groups.update(self.model.visibility_field.get_param_groups())
        return groups

    def generate_ddf_samples(self, randoms: Optional[Dict] = None) -> Dict[str, Any]:
        """:493-515"""
        if randoms is not None and "ddf_rays" in randoms:
            from ..cameras.rays import RayBundle
            o, d = randoms["ddf_rays"]
            dev = o.device
            rb = RayBundle(origins=o, directions=d, pixel_area=torch.ones(o.shape[0], 1, device=dev),
                           camera_indices=torch.zeros(o.shape[0], 1, dtype=torch.int64, device=dev),
                           metadata={"directions_norm": torch.ones(o.shape[0], 1, device=dev)})
        else:
            rb = self.visibility_train_sampler()
        data = self.model.generate_ddf_ground_truth(rb, self.config.visibility_accumulation_mask_threshold, randoms=randoms)
        data["sky_ray_bundle"] = randoms["sky_ray_bundle"] if randoms is not None and "sky_ray_bundle" in randoms else \
            self.datamanager.get_sky_ray_bundle(self.config.num_sky_rays)
        if self.config.stop_sdf_gradients:
            for k in ["accumulations", "mask", "termination_dist", "normals"]:
                data[k] = data[k].detach()
        return data

    # ------------------------------------------------------------------ gradients (the DDP wrapper's job, :198-199)
    def gradient_slab(self):
        """the flat gradient slab over every trainable parameter of get_param_groups() (groups in the optimizer-config order)"""
        if self._slab is None:
            from ..distributed import GradientSlab, slab_of
            groups = self.get_param_groups()
            self._slab = slab_of([p for ps in groups.values() for p in ps if p.requires_grad]) or GradientSlab(groups, self.world_size)
            self._slab.world_size = self.world_size
        return self._slab

    def exchange_with_one_rank(self) -> None:
        """run the exchange although world_size = 1 (a one-rank RCCL process group: bench.py's self-check and
        tests/test_gpu_trainer_surface.py -- the collective path on hardware when one GPU is all there is)"""
        self.exchange_gradients = True
        self.gradient_slab().force_exchange = True

    def _queue_exchange(self) -> None:
        if not self._exchange_queued:
            self._exchange_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)

    def _end_of_pass(self) -> None:
        """end of the trainer's backward pass (autograd final callback, on the stream `backward()` was called on, after the leaf
        streams were joined): collect -> all-reduce(mean); `p.grad` are the slab views when the trainer's optimizer runs"""
        from .. import ops
        self._exchange_queued = False
        slab = self.gradient_slab()
        ops.finish_pass()  # (its own callback may be queued behind this one)
        slab.collect(attach_unused=False)
        slab.all_reduce(self.config.bucketed_exchange)
        slab.wait()
        slab.exchanged = True

    def _with_gradient_exchange(self, loss_dict):
        """every differentiable scalar of the loss dict behind an identity node whose backward queues _end_of_pass (once per pass)"""
        from ..model_components.losses import LossDict
        ref = weakref.ref(self)
        wrap = lambda t: _ExchangeFn.apply(t, ref) if torch.is_tensor(t) and t.requires_grad else t  # noqa: E731
        out = LossDict({k: wrap(v) for k, v in loss_dict.items()})
        parts, total = getattr(loss_dict, "parts", None), getattr(loss_dict, "total", None)
        if parts:
            out.parts = [(wrap(x), c, sc) for x, c, sc in parts]
        if total is not None:
            out.total = wrap(total)
        return out

    def get_train_loss_dict(self, step: int, ray_bundle=None, batch=None, randoms: Optional[Dict] = None):
        """:241-291 -> (model_outputs, loss_dict, metrics_dict).  ray_bundle/batch/randoms may be injected (tests, bench); otherwise
        they come from the datamanager.  At world_size > 1 the backward pass of the returned losses ends with the gradient exchange;
        with config.graph_replay the step (and its backward) is a graph replay from call graph_replay_warmup + 1 on."""
        from .. import ops
        ops.reset_pass_state()  # (a backward pass that raised leaves its end-of-pass callbacks unrun)
        self._exchange_queued = False
        self._train_calls += 1
        if self.config.graph_replay and self.model.training and torch.is_grad_enabled() \
                and (self._train_graph is not None or self._train_calls > self.config.graph_replay_warmup):
            return self._replayed_train_loss_dict(step, ray_bundle, batch, randoms)
        exchange = self.exchange_gradients and self.model.training and torch.is_grad_enabled()
        if exchange:
            self.gradient_slab().begin_pass()
        model_outputs, loss_dict, metrics_dict = self._train_loss_dict(step, ray_bundle, batch, randoms)
        if exchange:
            loss_dict = self._with_gradient_exchange(loss_dict)
        return model_outputs, loss_dict, metrics_dict

    def _replayed_train_loss_dict(self, step: int, ray_bundle=None, batch=None, randoms: Optional[Dict] = None):
        sky = None
        if ray_bundle is None:
            ray_bundle, batch = self.datamanager.next_train(step)
        if randoms is None or "sky_ray_bundle" not in randoms:
            sky = self.datamanager.get_sky_ray_bundle(self.config.num_sky_rays)
        if self._train_graph is None:
            from .train_graph import TrainGraph
            if getattr(self.model, "illumination_shard", None) is not None:
                import torch.distributed as dist
                if dist.get_backend() != "nccl":
                    raise RuntimeError("graph_replay with shard_illumination_decode needs the RCCL backend: the colours' all-gather is part of the captured step")
            self._train_graph = TrainGraph(self, self.gradient_slab(), ray_bundle, batch, warmup=1, start_step=step, randoms=randoms)
        tg = self._train_graph
        tg.replay(step, ray_bundle, batch, sky, randoms)
        if self.exchange_gradients:
            tg.slab.all_reduce(self.config.bucketed_exchange)
            tg.slab.wait()
            tg.slab.exchanged = True
        return tg.outputs, tg.trainer_loss_dict(), tg.metrics

    def _train_loss_dict(self, step: int, ray_bundle=None, batch=None, randoms: Optional[Dict] = None):
        """the step itself (:241-291), eager: what get_train_loss_dict runs and what TrainGraph captures"""
        model = self.model
        if model.visibility_field is not None and not model.config.fit_visibility_field:
            model.visibility_field.eval()
        if ray_bundle is None:
            ray_bundle, batch = self.datamanager.next_train(step)
        model.set_step(step)
        model.begin_step()  # prepared-weight caches are per optimisation step
        fit = model.config.fit_visibility_field and model.visibility_field is not None
        fuse = fit and model.config.use_visibility and model.training
        vis_batch = None
        if model.training:
            model.start_illumination(ray_bundle, randoms=randoms)  # second stream; joined inside the model's forward
        if fuse:
            # The DDF-fit rays' ground truth only needs the SDF field, so it is produced first and the DDF evaluations
            # of the fit step are handed to compute_visibility to share its launches (same arithmetic as :271-289)
            vis_batch = self.generate_ddf_samples(randoms)
            prep = model.visibility_field.prepare_queries(vis_batch["ray_bundle"], vis_batch,
                                                          None if randoms is None else randoms.get("mv_points"))
            prep["stop_gradients"] = self.config.stop_sdf_gradients
            model._extra_ddf, model._extra_ddf_out = prep, None
            model.start_ddf_fit(prep)  # a small chain launch of its own, third stream; picked up inside the model's forward
        try:
            model_outputs = model(ray_bundle, batch=batch, step=step, randoms=randoms)
        finally:
            model._extra_ddf = None
        metrics_dict = model.get_metrics_dict(model_outputs, batch)
        loss_dict = model.get_loss_dict(model_outputs, batch, metrics_dict)
        if fit:
            if vis_batch is None:
                vis_batch = self.generate_ddf_samples(randoms)
            vis_outputs = model.visibility_field(ray_bundle=vis_batch["ray_bundle"], batch=vis_batch, neusky=model,
                                                 stop_gradients=self.config.stop_sdf_gradients,
                                                 mv_points=None if randoms is None else randoms.get("mv_points"),
                                                 precomputed=getattr(model, "_extra_ddf_out", None) if fuse else None)
            model._extra_ddf_out = None
            vis_metrics = model.visibility_field.get_metrics_dict(vis_outputs, vis_batch)
            vis_loss = model.visibility_field.get_loss_dict(vis_outputs, vis_batch, vis_metrics)
            model_outputs = {**model_outputs, **vis_outputs}
            loss_dict = merge_loss_dicts(loss_dict, vis_loss)
            metrics_dict = {**metrics_dict, **vis_metrics}
        return model_outputs, loss_dict, metrics_dict

    # ------------------------------------------------------------------ evaluation (neusky_pipeline.py:204-444)
    def _optimise_evaluation_latents(self, step) -> None:
        """:204-210: fit the eval illumination latents once per eval step"""
        if self.step_of_last_latent_optimisation != step:
            self.model.fit_latent_codes_for_eval(datamanager=self.datamanager, global_step=step)
            self.step_of_last_latent_optimisation = step

    def global_scale(self, pred_img: torch.Tensor, gt_img: torch.Tensor) -> torch.Tensor:
        """:212-225: least-squares optimal scalar alpha = <gt, pred> / <pred, pred>"""
        p, g = pred_img.reshape(-1), gt_img.reshape(-1).to(pred_img.device)
        return torch.dot(g, p) / torch.dot(p, p) * pred_img

    def _eval_mode(self, on: bool) -> None:
        m = self.model
        if on:
            m.eval()
            if m.visibility_field is not None:
                m.visibility_field.eval()
        else:
            m.train()
            if m.visibility_field is not None and m.config.fit_visibility_field:
                m.visibility_field.train()

    def get_eval_loss_dict(self, step: int):
        """:294-313"""
        self._optimise_evaluation_latents(step)
        self._eval_mode(True)
        try:
            ray_bundle, batch = self.datamanager.next_eval(step)
            with torch.no_grad():
                model_outputs = self.model(ray_bundle, step=step)
                metrics_dict = self.model.get_metrics_dict(model_outputs, batch)
                loss_dict = self.model.get_loss_dict(model_outputs, batch, metrics_dict)
        finally:
            self._eval_mode(False)
        return model_outputs, loss_dict, metrics_dict

    def get_eval_image_metrics_and_images(self, step: int):
        """:316-390.  The DDF depth-grid visualisation (:330-377: twelve look-at cameras on the sphere rendered through the DDF) is
        viewer output and is not produced (viewer / visualisation out of scope)."""
        self._optimise_evaluation_latents(step)
        self._eval_mode(True)
        try:
            self.eval_image_num = self.eval_image_num % self.max_eval_num
            image_idx, camera_ray_bundle, batch = self.datamanager.next_eval_image(self.eval_image_num)
            outputs = self.model.get_outputs_for_camera_ray_bundle(camera_ray_bundle, show_progress=True, step=step)
            if self.config.least_squares_global_scale:
                outputs["rgb"] = self.global_scale(outputs["rgb"], batch["image"])
            metrics_dict, images_dict = self.model.get_image_metrics_and_images(outputs, batch)
            assert "image_idx" not in metrics_dict
            metrics_dict["image_idx"] = image_idx
            assert "num_rays" not in metrics_dict
            metrics_dict["num_rays"] = int(camera_ray_bundle.origins.shape[0] * camera_ray_bundle.origins.shape[1]) \
                if camera_ray_bundle.origins.dim() == 3 else int(camera_ray_bundle.origins.shape[0])
        finally:
            self._eval_mode(False)
        self.eval_image_num += 1
        return metrics_dict, images_dict

    def get_average_eval_image_metrics(self, step: Optional[int] = None):
        """:393-444: every eval image once, metrics averaged; adds num_rays_per_sec and fps like the reference"""
        from time import time
        self._optimise_evaluation_latents(step)
        self._eval_mode(True)
        metrics_dict_list = []
        try:
            loader = getattr(self.datamanager, "eval_dataloader", None)
            num_images = len(loader.image_indices) if hasattr(loader, "image_indices") else (len(loader) if loader is not None else self.max_eval_num)
            for eval_image_num in range(num_images):
                image_idx, camera_ray_bundle, batch = self.datamanager.next_eval_image(eval_image_num)
                inner_start = time()
                height, width = camera_ray_bundle.origins.shape[:2]
                num_rays = height * width
                outputs = self.model.get_outputs_for_camera_ray_bundle(camera_ray_bundle, step=step)
                if self.config.least_squares_global_scale:
                    outputs["rgb"] = self.global_scale(outputs["rgb"], batch["image"])
                metrics_dict, _ = self.model.get_image_metrics_and_images(outputs, batch)
                torch.cuda.synchronize()
                assert "num_rays_per_sec" not in metrics_dict
                metrics_dict["num_rays_per_sec"] = num_rays / (time() - inner_start)
                assert "fps" not in metrics_dict
                metrics_dict["fps"] = metrics_dict["num_rays_per_sec"] / (height * width)
                metrics_dict_list.append(metrics_dict)
        finally:
            self._eval_mode(False)
        return {key: float(torch.mean(torch.tensor([float(m[key]) for m in metrics_dict_list]))) for key in metrics_dict_list[0].keys()}

    # ------------------------------------------------------------------ trainer hooks (nerfstudio VanillaPipeline surface)
    def get_training_callbacks(self, training_callback_attributes=None) -> List:
        """nerfstudio's Trainer asks the pipeline for per-iteration callbacks; the proposal-weight anneal is applied by
        `get_train_loss_dict` itself (model.set_step), so there is nothing to register."""
        return []

    def load_pipeline(self, loaded_state: Dict[str, Any], step: int) -> None:
        """nerfstudio Trainer._load_checkpoint: {"pipeline": state_dict} with or without the DDP `module.` prefix"""
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in loaded_state.items()}
        self.load_state_dict(state, strict=True)
        self.model.set_step(step)
        self.model.begin_step()

    @property
    def device(self):
        return self.model.device
