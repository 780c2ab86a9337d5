"""NeuSky model on the MI355X kernels.

Mirrors `neusky.models.neusky_model.NeuSkyFactoModel` (neusky/models/neusky_model.py:171-1778) for the
per-ray train / render step: `forward` (:425-443), `sample_illumination` (:445-551),
`sample_and_forward_field` (:553-736), `get_outputs` (:738-931), `get_loss_dict` (:933-1062, train branch),
`get_metrics_dict` (:1064-1077), `generate_ddf_ground_truth` (:1337-1367),
`get_outputs_for_camera_ray_bundle` (:1369-1501), `compute_visibility` (:1624-1778), `get_param_groups`
(:379-398).  Viewer GUI, animation and image-metric code are out of scope.

Differences in kind, not in arithmetic: the reference's [R*S, D, *] broadcasts of light directions,
light colours and visibility are never built (compact [D,3] / [U,D,3] + camera row / [R,D] tensors go
straight into the fused hemisphere kernel); every random draw of a step can be injected (`randoms`) so the
step is reproducible against the CPU oracle.
"""
from __future__ import annotations

import os

import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple, Type, Union

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Parameter

from .. import hip, ops
from ..cameras.rays import Frustums, RayBundle, RaySamples
from ..field_components.neusky_fieldheadnames import FieldHeadNames, NeuSkyFieldHeadNames
from ..fields.sdf_albedo_field import SDFAlbedoFieldConfig
from ..model_components.illumination import IcosahedronSamplerConfig, RENIFieldConfig
from ..model_components.losses import (LossDict, RENISkyPixelLoss, interlevel_loss, interlevel_per_ray, monosdf_normal_loss, scale_dict,
                                        total_loss)
from ..model_components.ray_samplers import HashMLPDensityField, ProposalNetworkSampler
from ..model_components.renderers import RGBLambertianRendererWithVisibility
from ..utils.utils import device_rng, device_rng_seed, linear_to_sRGB, to_device_async
from ..plugin import ConfigBase, ModelBase


def _default_loss_inclusions() -> Dict[str, Any]:  # neusky/configs/neusky_config.py:102-126
    return {
        "rgb_l1_loss": True, "rgb_l2_loss": False, "cosine_colour_loss": False, "eikonal loss": True, "fg_mask_loss": True,
        "normal_loss": False, "depth_loss": False, "sdf_level_set_visibility_loss": True, "interlevel_loss": True,
        "sky_pixel_loss": {"enabled": True, "cosine_weight": 0.1},
        "hashgrid_density_loss": {"enabled": True, "grid_resolution": 10},
        "ground_plane_loss": True,
        "visibility_sigmoid_loss": {"visibility_threshold_method": "learnable", "optimise_sigmoid_bias": True,
                                    "optimise_sigmoid_scale": False, "target_min_bias": 0.1, "target_max_scale": 25,
                                    "steps_until_min_bias": 50000},
    }


_COEF_VECTORS: Dict[tuple, torch.Tensor] = {}


def _default_loss_coefficients() -> Dict[str, float]:  # neusky/configs/neusky_config.py:127-141
    return {"rgb_l1_loss": 1.0, "rgb_l2_loss": 0.0, "cosine_colour_loss": 1.0, "eikonal loss": 0.1, "fg_mask_loss": 1.0,
            "normal_loss": 1.0, "depth_loss": 1.0, "sdf_level_set_visibility_loss": 1.0, "interlevel_loss": 1.0,
            "sky_pixel_loss": 1.0, "hashgrid_density_loss": 1e-4, "ground_plane_loss": 0.1, "visibility_sigmoid_loss": 0.01}


@dataclass
class NeuSkyFactoModelConfig(ConfigBase):
    """neusky/models/neusky_model.py:81-169 + inherited NeuSFactoModelConfig members,
    defaults = the `neusky` method (neusky/configs/neusky_config.py:65-161)."""

    _target: Type = field(default_factory=lambda: NeuSkyFactoModel)
    sdf_field: SDFAlbedoFieldConfig = field(default_factory=SDFAlbedoFieldConfig)
    illumination_field: RENIFieldConfig = field(default_factory=RENIFieldConfig)
    illumination_sampler: IcosahedronSamplerConfig = field(default_factory=IcosahedronSamplerConfig)
    num_proposal_samples_per_ray: Tuple[int, ...] = (256, 96)
    num_neus_samples_per_ray: int = 48
    num_proposal_iterations: int = 2
    proposal_net_args_list: List[Dict] = field(default_factory=lambda: [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 64},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 256}])
    use_proposal_weight_anneal: bool = True
    proposal_weights_anneal_slope: float = 10.0
    proposal_weights_anneal_max_num_iters: int = 1000
    use_single_jitter: bool = True
    loss_inclusions: Dict[str, Any] = field(default_factory=_default_loss_inclusions)
    loss_coefficients: Dict[str, float] = field(default_factory=_default_loss_coefficients)
    use_visibility: bool = True
    fit_visibility_field: bool = True
    sdf_to_visibility_stop_gradients: str = "depth"
    only_upperhemisphere_visibility: bool = True
    lower_hermisphere_visibility: bool = True
    fix_test_illumination_directions: bool = True
    eval_num_rays_per_chunk: int = 256
    scene_contraction_order: str = "L2"
    collider_shape: str = "sphere"
    near_plane: float = 0.05
    visibility_threshold: Union[str, float] = "learnable"
    render_ambient_light: bool = False
    # eval-latent fitting (neusky_model.py:151-168, values of neusky_config.py:142-149)
    eval_latent_optimise_method: str = "per_image"      # per_image | nerf_osr_holdout | nerf_osr_envmap
    eval_latent_sample_region: str = "full_image"       # left_image_half | right_image_half | full_image
    optimise_compare_eval_scale: bool = False
    eval_latent_optimizer: Dict[str, Any] = field(default_factory=lambda: {"lr": 1e-1, "eps": 1e-15, "lr_final": 1e-7, "max_steps": 250})

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class NeuSkyFactoModel(ModelBase):
    config: NeuSkyFactoModelConfig

    def __init__(self, config: NeuSkyFactoModelConfig, scene_box, num_train_data: int, num_val_data: int, num_test_data: int,
                 visibility_field, test_mode: str, **kwargs) -> None:
        nn.Module.__init__(self)  # not the nerfstudio base's constructor (neusky_amd/plugin.py)
        self.config = config
        self.scene_box = scene_box
        self.num_train_data, self.num_val_data, self.num_test_data = num_train_data, num_val_data, num_test_data
        self.test_mode = test_mode
        self.num_eval_data = num_val_data if test_mode == "val" else num_test_data
        self.fitting_eval_latents = False
        self.train_metadata = kwargs.get("train_metadata", None)
        self.eval_metadata = kwargs.get("eval_metadata", None)
        if config.collider_shape != "sphere":
            raise NotImplementedError("the neusky config uses the unit-sphere collider (neusky_config.py:159)")
        self.populate_modules()
        self.visibility_field = visibility_field  # neusky_model.py:203 (registered as a sub-module on purpose)
        if self.visibility_field is not None:  # :217-246
            self.ddf_radius = self.visibility_field.ddf_radius
            vsl = config.loss_inclusions["visibility_sigmoid_loss"]
            self.visibility_threshold_method = vsl["visibility_threshold_method"]
            if self.visibility_threshold_method != "learnable" or vsl["optimise_sigmoid_scale"] or not vsl["optimise_sigmoid_bias"]:
                raise NotImplementedError("only the learnable-bias / fixed-scale visibility sigmoid of neusky_config.py:118-125")
            self.sigmoid_scale = float(vsl["target_max_scale"])
            self.visibility_threshold = Parameter(torch.tensor([self.ddf_radius * 2.0]))  # :234

    # ------------------------------------------------------------------ construction
    def populate_modules(self) -> None:
        c = self.config
        aabb = self.scene_box["aabb"] if isinstance(self.scene_box, dict) else self.scene_box.aabb
        self.field = c.sdf_field.setup(aabb=aabb, num_images=self.num_train_data, spatial_distortion=None)
        self.proposal_networks = nn.ModuleList([
            HashMLPDensityField(hidden_dim=a["hidden_dim"], num_levels=a["num_levels"], max_res=a["max_res"],
                                log2_hashmap_size=a["log2_hashmap_size"]) for a in c.proposal_net_args_list])
        self.density_fns = [n.density_fn for n in self.proposal_networks]
        self.proposal_sampler = ProposalNetworkSampler(
            num_nerf_samples_per_ray=c.num_neus_samples_per_ray, num_proposal_samples_per_ray=c.num_proposal_samples_per_ray,
            num_proposal_network_iterations=c.num_proposal_iterations)
        # illumination (neusky_model.py:253-300); decoder frozen, per-image latents + scale trainable
        self.illumination_field = c.illumination_field.setup(num_train_data=None, num_eval_data=None)
        L = self.illumination_field.latent_dim
        self.train_illumination_latents = Parameter(torch.zeros((self.num_train_data, L, 3)))
        self.train_scale = Parameter(torch.ones(self.num_train_data))
        self.eval_illumination_latents = Parameter(torch.zeros((max(self.num_eval_data, 1), L, 3)))
        self.eval_scale = Parameter(torch.ones(max(self.num_eval_data, 1)))
        self.eval_rotation = Parameter(torch.ones(max(self.num_eval_data, 1)))  # :259 (nerf_osr_envmap: z-rotation per eval session)
        self.illumination_sampler = c.illumination_sampler.setup()
        self.lambertian_renderer = RGBLambertianRendererWithVisibility()
        li = c.loss_inclusions
        assert not (li["rgb_l1_loss"] and li["rgb_l2_loss"]), "Cannot have both L1 and L2 loss"  # :359-361
        if li["sky_pixel_loss"]["enabled"]:
            self.sky_pixel_loss = RENISkyPixelLoss(alpha=li["sky_pixel_loss"]["cosine_weight"])
        self._step = 0

    @property
    def device(self):
        return self.visibility_threshold.device if hasattr(self, "visibility_threshold") else next(self.parameters()).device

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        """neusky_model.py:379-398"""
        groups = {
            "fields": list(self.field.parameters()),
            "proposal_networks": list(self.proposal_networks.parameters()),
            "illumination_field": [self.train_illumination_latents, self.train_scale],
        }
        if self.visibility_field is not None:
            groups["visibility_sigmoid"] = [self.visibility_threshold]
        return groups

    def get_illumination_field(self):
        """neusky_model.py:400-412"""
        if self.training and not self.fitting_eval_latents:
            return self.train_illumination_latents, self.train_scale
        return self.eval_illumination_latents, self.eval_scale

    def set_step(self, step: int) -> None:
        """proposal-weight annealing callback of NeuSFacto/nerfacto"""
        self._step = step
        c = self.config
        if c.use_proposal_weight_anneal:
            N = c.proposal_weights_anneal_max_num_iters
            frac = min(max(step / N, 0.0), 1.0)
            bias = lambda x, b: (b * x) / ((b - 1) * x + 1)
            self.proposal_sampler.set_anneal(bias(frac, c.proposal_weights_anneal_slope))

    def begin_step(self) -> None:
        """start of an optimisation step: drop the per-step caches of prepared (weight-normed / padded) matrices"""
        ops.begin_step(self.device)
        self.field.invalidate_weight_cache()
        for net in self.proposal_networks:
            net.invalidate_weight_cache()
        self.illumination_field.network.invalidate_weight_cache()
        if self.visibility_field is not None:
            self.visibility_field.field.ddf.invalidate_weight_cache()

    # ------------------------------------------------------------------ collider
    def collider(self, ray_bundle: RayBundle) -> RayBundle:
        """nerfstudio SphereCollider(center=0, radius=1, near_plane=0.05) (neusky_model.py:213)"""
        nears, fars = hip.sphere_collider(ray_bundle.origins.detach().contiguous(), ray_bundle.directions.detach().contiguous(),
                                          1.0, float(self.config.near_plane))
        ray_bundle.nears, ray_bundle.fars = nears, fars
        return ray_bundle

    def forward(self, ray_bundle: RayBundle, batch: Optional[Dict] = None, rotation: Optional[torch.Tensor] = None,
                step: Optional[int] = None, randoms: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, Any]:
        """neusky_model.py:425-443"""
        ray_bundle = self.collider(ray_bundle)
        if not self.training:
            self.begin_step()  # no optimiser between eval forwards, but parameters may have been loaded / fitted
        return self.get_outputs(ray_bundle, batch=batch, rotation=rotation, step=step, randoms=randoms)

    # ------------------------------------------------------------------ sampling + field
    def _sample(self, ray_bundle: RayBundle, randoms: Optional[Dict], want_inds: bool = False):
        R = ray_bundle.origins.shape[0]
        dev = ray_bundle.origins.device
        n_lvls = self.config.num_proposal_iterations + 1
        if self.training:
            if randoms is not None and "jitters" in randoms:
                jitters = [j.to(dev) for j in randoms["jitters"]]
            else:
                jitters = list(torch.rand(n_lvls, R, 1, device=dev).unbind(0))  # one draw for all levels
        else:
            jitters = None
        sbins, ebins, weights_list, sbins_list, inds_list = self.proposal_sampler(
            ray_bundle.origins, ray_bundle.directions, ray_bundle.nears, ray_bundle.fars, self.density_fns, jitters, want_inds)
        S = ebins.shape[1] - 1
        o = ray_bundle.origins[:, None, :].expand(R, S, 3)
        d = ray_bundle.directions[:, None, :].expand(R, S, 3)
        cam = ray_bundle.camera_indices.reshape(R, 1, 1).expand(R, S, 1) if ray_bundle.camera_indices is not None else None
        # starts and ends as two CONTIGUOUS [R, S] matrices from one launch: as column slices of the bins every consumer kernel
        # (NeuS weights, the per-ray reductions, twice each per step) first copied them
        se = torch.stack((ebins[:, :-1], ebins[:, 1:]))
        rs = RaySamples(frustums=Frustums(origins=o, directions=d, starts=se[0][..., None], ends=se[1][..., None],
                                          pixel_area=None),
                        camera_indices=cam, deltas=(se[1] - se[0])[..., None],
                        spacing_starts=sbins[:, :-1, None], spacing_ends=sbins[:, 1:, None])
        return rs, weights_list, sbins_list, sbins, inds_list

    def sample_illumination(self, ray_samples: RaySamples, rotation: Optional[torch.Tensor] = None):
        """neusky_model.py:445-551 with the reference's signature and return layout: (hdr illumination colours [R*S, D, 3],
        illumination directions [R*S, D, 3], hdr background colours [R, 3]).  The broadcast tensors are materialised HERE
        only, for callers written against the reference; the step itself runs sample_illumination_compact."""
        cam = ray_samples.camera_indices[:, 0, 0] if ray_samples.camera_indices.dim() == 3 else ray_samples.camera_indices.reshape(-1)
        R, S = ray_samples.frustums.origins.shape[:2]
        dirs, cols, cam_of_ray, bg = self.sample_illumination_compact(cam, ray_samples.frustums.directions[:, 0].contiguous(), rotation)
        D = dirs.shape[0]
        colours = cols[cam_of_ray.long()][:, None].expand(R, S, D, 3).reshape(R * S, D, 3)
        return colours, dirs[None].expand(R * S, D, 3), bg

    def sample_illumination_compact(self, camera_indices: torch.Tensor, ray_directions: torch.Tensor,
                                    rotation: Optional[torch.Tensor] = None, randoms: Optional[Dict] = None):
        """neusky_model.py:445-551 on compact data: camera_indices [R], ray_directions [R,3] ->
        directions [D,3], cam_colours [U,D,3], cam_of_ray [R] (int32 row of cam_colours), hdr_background [R,3]."""
        latents, scales = self.get_illumination_field()
        frame = getattr(self, "_frame_illumination", None)
        if frame is not None and not self.training:
            # chunked full-frame render: directions and the frame camera's colours were decoded once per frame
            dirs, cols, sel, cam, rot_f = frame
            self._upper_sel = sel
            R = camera_indices.shape[0]
            bg = self.illumination_field.forward_camera(ray_directions, latents[cam], scales[cam], rot_f)
            return dirs, cols, torch.zeros(R, dtype=torch.int32, device=dirs.device), bg
        if not self.training and self.config.fix_test_illumination_directions:
            dirs, sel = self.illumination_sampler.on_device(self.device, apply_random_rotation=False)  # :451-454
        elif randoms is not None and "light_rotation" in randoms:
            dirs, sel = self.illumination_sampler.on_device(self.device, rotation=randoms["light_rotation"])
        else:
            dirs, sel = self.illumination_sampler.on_device(self.device)  # :456-458, drawn on the device
        self._upper_sel = sel  # upper-hemisphere subset (:1650-1657): static size D/2 for the antipodal direction set
        D = dirs.shape[0]
        if (self.training or self.fitting_eval_latents) and latents.shape[0] <= max(1024, camera_indices.shape[0]):
            # every camera of the active latent set is decoded (U = num_train_data, or num_eval_data while the eval latents
            # are being fitted: static shape, no torch.unique host sync, hipGraph-safe); rows of cameras absent from the
            # batch are never read by the renderer and receive zero gradient
            inverse = camera_indices
            shard = getattr(self, "illumination_shard", None) if self.training else None
            if shard is not None and rotation is None:
                # camera-sharded decode (distributed.CameraAllGather): this rank decodes cameras r::N, the colours are all-gathered,
                # the gradient w.r.t. them reduce-scattered; the rays' own background rows are this rank's rays: decoded here
                from ..distributed import CameraAllGather
                r, n = shard
                own = self.illumination_field.forward_grid(dirs, latents[r::n].contiguous(), scales[r::n].contiguous())
                cols = CameraAllGather.apply(own, latents.shape[0], r, n)
                bg = self.illumination_field(ray_directions, latents[camera_indices], scales[camera_indices], None)
                return dirs, cols, inverse.to(torch.int32), bg
            if rotation is None:  # the rays' own background rows (:535-549) ride in the same decoder pass
                cols, bg = self.illumination_field.forward_grid_and_rays(dirs, latents, scales, ray_directions, camera_indices)
                return dirs, cols, inverse.to(torch.int32), bg
            cols = None
            unique = torch.arange(latents.shape[0], device=dirs.device)
        else:
            unique, inverse = torch.unique(camera_indices, return_inverse=True)  # :461-463
            cols = None
        U = unique.shape[0]
        if cols is not None:
            pass
        elif rotation is None:
            cols = self.illumination_field.forward_grid(dirs, latents[unique], scales[unique])  # :488-510, no per-pair gather
        else:
            ci = unique[:, None].expand(U, D).reshape(-1)
            dd = dirs[None].expand(U, D, 3).reshape(-1, 3)
            rot = rotation if rotation.dim() == 2 else rotation[ci]
            cols = self.illumination_field(dd, latents[ci], scales[ci], rot).reshape(U, D, 3)
        rot_r = rotation if (rotation is None or rotation.dim() == 2) else rotation[camera_indices]
        bg = self.illumination_field(ray_directions, latents[camera_indices], scales[camera_indices], rot_r)  # :535-549
        return dirs, cols, inverse.to(torch.int32), bg

    def start_illumination(self, ray_bundle: RayBundle, rotation=None, randoms=None) -> None:
        """Launch the step's illumination decode on the second stream NOW (it depends on nothing but the batch's camera
        indices, ray directions and the latents); sample_and_forward_field picks the result up.  The pipeline calls this
        before the DDF-fit ground-truth pass, whose sampler geometry is a run of ~180 small launches that leave the chip
        idle, so the decode's dense layers fill it."""
        cam = ray_bundle.camera_indices.reshape(-1)
        if not (self.training and self.second_stream):
            return
        main = torch.cuda.current_stream()
        side = self._illumination_stream()
        if side == main:  # (the caller's stream IS this pool stream: sample_and_forward_field decodes in line)
            return
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._illumination_pending = self.sample_illumination_compact(cam, ray_bundle.directions, rotation, randoms)

    def start_ddf_fit(self, prep: Dict[str, Any]) -> None:
        """Evaluate the DDF on the step's fit rows (fit rays | multi-view | sky: DDFModel.prepare_queries) NOW, as a small chain launch
        of its own on a third stream; compute_visibility_compact picks the result up.  In the same launch as the 262 144 visibility
        rows (rounds 3-5) they were the 11 workgroups of a ninth round on 11 of 256 CUs; here they run beside the proposal sampler and
        the field pass of the main rays, and their backward beside the model's.  The hash encode of their sphere points stays on the
        CALLER's stream: its backward scatters into the DDF table, and the big scatter of the visibility rows writes its chunks back
        without atomics -- the two must stay ordered."""
        ddf = self.visibility_field
        fork = self.second_stream and ops.ASYNC_WGRAD  # (in line: the small node's weight gradients share accumulators with the big one's)
        pts, xrow, mv_points, sky_gt, dist_w = ops.DDFFitRowsFn.apply(prep["term_dist"], self.ddf_radius, prep)
        cond = ddf.field.condition_rows(pts)
        main = torch.cuda.current_stream()
        side = self._ddf_fit_stream() if fork else main
        fork = fork and side != main
        if fork:
            side.wait_stream(main)
        N, Ns = prep["positions"].shape[0], (prep["sky_o"].shape[0] if prep["sky_o"] is not None else 0)
        with torch.cuda.stream(side):
            t = ddf.field.forward_encoded(xrow, cond)
            # (the split too: autograd runs a node's backward on its forward's stream, and the split's backward -- enqueued late, it is
            # one of the first nodes of the step -- would otherwise sit at the end of the MAIN stream's queue in front of the small chain)
            t_main, t_mv, t_sky = torch.split(t, [N, mv_points.shape[0], Ns])
        # (held until the next step replaces it: memory of the side stream's pool that this stream reads)
        self._extra_ddf_eval = {"mv_points": mv_points, "t_main": t_main, "t_mv": t_mv, "t_sky": t_sky, "sky_gt": sky_gt,
                                "distance_weight": dist_w, "stream": side if fork else None, "all": (t, cond, xrow, pts)}

    def _ddf_fit_stream(self):
        s = getattr(self, "_fit_stream", None)
        if s is None:
            s = self._fit_stream = ops.role_stream("ddf_fit", self.device)  # (one per process and device: ops.role_stream)
        return s

    # False: the illumination decode runs in line on the caller's stream (bench.py's per-kernel timing iteration, where a kernel
    # sharing the chip with the other stream's work would be timed with that work's share of the CUs missing)
    second_stream: bool = True

    def _illumination_stream(self):
        s = getattr(self, "_illum_stream", None)
        if s is None:
            s = self._illum_stream = ops.role_stream("illumination", self.device)  # (one per process and device: ops.role_stream)
        return s

    def render_depth(self, weights: torch.Tensor, ray_samples: RaySamples) -> torch.Tensor:
        """nerfstudio DepthRenderer('expected') (neusky_model.py:591): weights [R,S,1] -> [R,1]"""
        return self.ray_reductions(weights, ray_samples)[0]

    @staticmethod
    def ray_reductions(weights: torch.Tensor, ray_samples: RaySamples, normals: Optional[torch.Tensor] = None,
                       albedo: Optional[torch.Tensor] = None, max_clamp: float = 0.0):
        """(p2p_dist [R,1], accumulation [R,1], normal [R,3], albedo-on-white [R,3]) of :591-595 / :812-813 / :1342-1357 from one
        fused pass (ops.RayReduceFn)"""
        fr = ray_samples.frustums
        return ops.RayReduceFn.apply(weights, fr.starts, fr.ends, normals, albedo, float(max_clamp))

    def compute_visibility(self, ray_samples: RaySamples, depth: torch.Tensor, illumination_directions: torch.Tensor,
                           threshold_distance: torch.Tensor, sigmoid_scale: float, compute_shadow_map: bool = False) -> Dict[str, Any]:
        """neusky_model.py:1624-1778 with the reference's signature: ray_samples [R,S] (sample 0 of each ray is used, :1667-1668),
        illumination_directions [R*S, D, 3] (row 0 is used, :1648) or [D, 3]; `visibility` comes back in the reference layout
        [R*S, D, 1] (repeated over the samples, :1755-1759).  The step itself runs compute_visibility_compact ([R, D])."""
        fr = ray_samples.frustums
        R, S = fr.origins.shape[:2]
        dirs = illumination_directions[0] if illumination_directions.dim() == 3 else illumination_directions
        out = self.compute_visibility_compact(fr.origins[:, 0].contiguous(), fr.directions[:, 0].contiguous(), depth, dirs.contiguous(),
                                              threshold_distance, sigmoid_scale, compute_shadow_map)
        D = dirs.shape[0]
        out["visibility"] = out["visibility"][:, None, :, None].expand(R, S, D, 1).reshape(R * S, D, 1)
        return out

    def compute_visibility_compact(self, origins: torch.Tensor, ray_directions: torch.Tensor, depth: torch.Tensor,
                                   illumination_directions: torch.Tensor, threshold_distance: torch.Tensor, sigmoid_scale: float,
                                   compute_shadow_map: bool = False, sel: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        """neusky_model.py:1624-1778 on compact data: origins / ray_directions [R,3] (= sample 0 of each ray,
        :1667-1668), depth [R,1], illumination_directions [D,3] (= row 0 of the broadcast, :1648).
        Returns visibility [R,D] (the reference repeats it over S, :1755-1759)."""
        R, D = origins.shape[0], illumination_directions.shape[0]
        dev = origins.device
        if not self.config.only_upperhemisphere_visibility:
            sel = torch.arange(D, device=dev, dtype=torch.int32)
        elif sel is not None:
            pass  # chosen on the host by sample_illumination
        else:  # :1650-1657
            sel = torch.nonzero(illumination_directions[:, 2] > 0)[:, 0].to(torch.int32)
        Dv = sel.numel()
        sel_dirs = illumination_directions[sel.long()].contiguous()
        M = R * Dv
        ddf = self.visibility_field
        extra = getattr(self, "_extra_ddf", None) if self.training else None
        # the DDF-fit rows (rays | multi-view | sky, ddf_model.py:217,319,360) were evaluated by start_ddf_fit (a small launch of their own
        # on a third stream: the visibility rows are exactly 8 rounds of the chain kernels); without it they ride behind the visibility
        # rows in the same buffers and launches (ops.DDFQueryRowsFn)
        ev = getattr(self, "_extra_ddf_eval", None) if extra is not None else None
        self._extra_ddf_eval = None
        ride = extra if ev is None else None
        pts_all, xrow_all, surf_dist, term_dist, mv_points, sky_gt, dist_w = ops.DDFQueryRowsFn.apply(
            None if ride is None else ride["term_dist"], origins.detach().contiguous(), ray_directions.detach().contiguous(),
            depth.detach().reshape(-1).contiguous(), sel_dirs, self.ddf_radius, ride)
        t_all = ddf.field.forward_rows(pts_all, xrow_all)  # :1716 -> ddf_model.py:217
        if ev is not None:
            if ev["stream"] is not None:
                torch.cuda.current_stream().wait_stream(ev["stream"])
            # the side stream's tensors stay referenced until the next step's replace them -- their MEMORY, not their autograd graph:
            # a graph kept past its step keeps the parameters' AccumulateGrad nodes, which remember the stream they were made on
            self._extra_ddf_keep = [t.detach() for t in ev["all"]]
            t_hat, t_main, t_mv, t_sky = t_all, ev["t_main"], ev["t_mv"], ev["t_sky"]
            mv_points, sky_gt, dist_w = ev["mv_points"], ev["sky_gt"], ev["distance_weight"]
        elif extra is not None:
            N, Ns = extra["positions"].shape[0], (extra["sky_o"].shape[0] if extra["sky_o"] is not None else 0)
            t_hat, t_main, t_mv, t_sky = torch.split(t_all, [M, N, mv_points.shape[0], Ns])
        else:
            t_hat = t_all
        sphere_pts = pts_all[:M]
        out: Dict[str, Any] = {"expected_termination_dist": t_hat}
        stop_gradients = self.config.sdf_to_visibility_stop_gradients in ["sdf", "both"]  # :1712-1714
        vcfg = ddf.config
        sdf_extra = None
        if (vcfg.loss_inclusions["sdf_l1_loss"] or vcfg.loss_inclusions["sdf_l2_loss"]) and ddf.training:
            n_main = 0
            if extra is not None and not stop_gradients and not extra["stop_gradients"]:
                n_main = extra["positions"].shape[0]  # the fit rays' own termination points (ddf_model.py:243) join the same probe
            # sphere point - direction x predicted distance (ddf_model.py:243), one kernel per row set into one buffer
            term_pts = ops.TermPointsFn.apply(sphere_pts, sel_dirs, t_hat, extra["positions"] if n_main else None,
                                              extra["directions"] if n_main else None, t_main if n_main else None)
            if stop_gradients:
                with torch.no_grad():
                    sdf_all = self.field.get_sdf_at_pos(term_pts).detach()
            else:
                sdf_all = self.field.get_sdf_at_pos(term_pts)
            out["sdf_at_termination"] = sdf_all[:M]
            if n_main:
                sdf_extra = sdf_all[M:]
        if extra is not None:
            self._extra_ddf_out = {"mv_points": mv_points, "t_main": t_main, "t_mv": t_mv, "t_sky": t_sky, "sdf_main": sdf_extra,
                                   "sky_gt": sky_gt, "distance_weight": dist_w}
        lower = 1.0 if self.config.lower_hermisphere_visibility else 0.0
        vis = ops.VisibilityFinishFn.apply(t_hat, surf_dist, threshold_distance, float(sigmoid_scale),
                                           sel.contiguous(), R, Dv, D, lower)
        out["visibility"] = vis
        if compute_shadow_map:
            out["difference"] = surf_dist - t_hat
        out["visibility_batch"] = {"termination_dist": term_dist, "mask": torch.ones_like(term_dist),
                                   "sdf_at_termination": out.get("sdf_at_termination")}
        return out

    def sample_and_forward_field(self, ray_bundle: RayBundle, batch=None, rotation=None, step=None, randoms=None) -> Dict[str, Any]:
        """neusky_model.py:553-736"""
        cam = ray_bundle.camera_indices.reshape(-1)
        # The illumination decode (big dense layers, nothing but the camera indices as input) runs on a second HIP stream
        # beside the proposal sampler + field pass (hundreds of small launches): a fork/join that the HIP graph keeps as
        # two parallel branches, forward and backward (autograd replays each node on its forward stream).
        fork = self.training and self.second_stream
        pending = getattr(self, "_illumination_pending", None)
        self._illumination_pending = None
        if fork and pending is not None:  # started by start_illumination (the pipeline, before the DDF-fit ground truth pass)
            main, side = torch.cuda.current_stream(), self._illumination_stream()
            dirs, cam_colours, cam_of_ray, hdr_bg = pending
        elif fork and self._illumination_stream() == torch.cuda.current_stream():
            fork = False  # (the caller's stream IS the illumination pool stream: in line, below)
        elif fork:
            main = torch.cuda.current_stream()
            side = self._illumination_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dirs, cam_colours, cam_of_ray, hdr_bg = self.sample_illumination_compact(cam, ray_bundle.directions, rotation, randoms)
        ray_samples, weights_list, sbins_list, sbins, inds_list = self._sample(ray_bundle, randoms, want_inds=randoms is not None)
        probe = self._grid_probe_points(ray_bundle.origins.device, randoms)
        field_outputs = self.field(ray_samples, return_alphas=True, extra_points=None if probe is None else probe[0])
        weights = field_outputs["weights"]
        weights_list = weights_list + [weights[..., 0]]
        sbins_list = sbins_list + [sbins]
        if fork:
            main.wait_stream(side)
            for t in (dirs, cam_colours, cam_of_ray, hdr_bg, getattr(self, "_upper_sel", None)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(main)  # allocated on the side stream, consumed on this one from here on
        else:
            dirs, cam_colours, cam_of_ray, hdr_bg = self.sample_illumination_compact(cam, ray_bundle.directions, rotation, randoms)
        out: Dict[str, Any] = {
            "ray_samples": ray_samples, "field_outputs": field_outputs, "weights": weights,
            "bg_transmittance": field_outputs["bg_transmittance"], "weights_list": weights_list, "sbins_list": sbins_list,
            "pdf_inds_list": inds_list, "illumination_directions": dirs, "hdr_illumination_colours": cam_colours,
            "cam_of_ray": cam_of_ray, "hdr_background_colours": hdr_bg,
        }
        # :591, :595 and get_outputs' :812-813 in one pass
        p2p_dist, accumulation, out["normal"], out["albedo_on_white"] = self.ray_reductions(
            weights, ray_samples, field_outputs[FieldHeadNames.NORMALS], field_outputs[NeuSkyFieldHeadNames.ALBEDO])
        out.update(p2p_dist=p2p_dist, accumulation=accumulation)
        if self.config.use_visibility:
            depth = p2p_dist / ray_bundle.metadata["directions_norm"]  # :593
            p2p_vis = p2p_dist.detach() if self.config.sdf_to_visibility_stop_gradients in ["depth", "both"] else p2p_dist
            if p2p_vis.requires_grad:
                raise NotImplementedError("visibility geometry is differentiated only in 'depth'/'both' mode (neusky_config.py:156)")
            out["visibility_dict"] = self.compute_visibility_compact(ray_bundle.origins, ray_bundle.directions, p2p_vis, dirs,
                                                             self.visibility_threshold, self.sigmoid_scale,
                                                             sel=getattr(self, "_upper_sel", None))
            out["depth"] = depth
        if probe is not None:
            # (sic) the reference hands `deltas=gap` ([3]) to get_alpha, which broadcasts [P,1]*[3] -> three alphas
            # per point, one per axis gap (equal for the cubic scene box) (:715-724, :732).  The probe points rode through the field
            # behind the batch's own samples (one pass instead of a second one over 1000 points: every kernel of a pass costs its
            # 25-40 us whatever the row count); their sdf / gradients come back as the tail rows.
            out["grid_density"] = ops.PointAlphasFn.apply(field_outputs["extra_sdf"], field_outputs["extra_gradients"], probe[1], self._grid_gap_host,
                                                          self.field.deviation_network.variance, self.field._cos_anneal_ratio)
        return out

    def _grid_probe_points(self, dev, randoms):
        """the hash-grid density probe's points and directions (:672-724), or None when the loss is off"""
        if not (self.training and self.config.loss_inclusions["hashgrid_density_loss"]["enabled"]):
            return None
        res = self.config.loss_inclusions["hashgrid_density_loss"]["grid_resolution"]
        aabb = self.scene_box["aabb"] if isinstance(self.scene_box, dict) else self.scene_box.aabb
        mn, mx = aabb[0], aabb[1]
        key = (res, str(dev))
        if getattr(self, "_grid_cache_key", None) != key:  # lattice + gaps built once, kept on the device
            lin = [torch.linspace(float(mn[i]), float(mx[i]), res) for i in range(3)]
            X, Y, Z = torch.meshgrid(*lin, indexing="ij")
            self._grid_lattice = torch.stack((X, Y, Z), -1).reshape(-1, 3).to(dev)
            self._grid_gap_host = [(float(mx[i]) - float(mn[i])) / res for i in range(3)]
            self._grid_gap = torch.tensor(self._grid_gap_host).to(dev)
            self._grid_cache_key = key
        gap = self._grid_gap
        if randoms is not None and "grid_perturb" in randoms:
            perturb, gdir = randoms["grid_perturb"].to(dev), randoms["grid_dirs"].to(dev)
            positions = self._grid_lattice + (perturb * gap - gap / 2)
            gdir = gdir / torch.norm(gdir, dim=-1, keepdim=True)
        else:  # the reference draws these on the CPU every step (:704-712); drawn in one kernel here (csrc/samplers.hip)
            positions, gdir = torch.empty_like(self._grid_lattice), torch.empty_like(self._grid_lattice)
            g_seed, g_counter = device_rng(self, "grid_probe_points", 3, dev)
            hip.grid_probe_points(self._grid_lattice, self._grid_gap_host, g_seed, g_counter, positions, gdir)
        return positions, gdir

    # ------------------------------------------------------------------ outputs
    def get_outputs(self, ray_bundle: RayBundle, batch=None, rotation=None, step=None, randoms=None) -> Dict[str, Any]:
        """neusky_model.py:738-931"""
        so = self.sample_and_forward_field(ray_bundle, batch=batch, rotation=rotation, step=step, randoms=randoms)
        fo = so["field_outputs"]
        weights, ray_samples = so["weights"], so["ray_samples"]
        visibility = so["visibility_dict"]["visibility"] if self.config.use_visibility else None
        sdf_at_termination = so["visibility_dict"].get("sdf_at_termination") if self.config.use_visibility else None
        rgb = self.lambertian_renderer.forward_compact(
            albedos=fo[NeuSkyFieldHeadNames.ALBEDO], normals=fo[FieldHeadNames.NORMALS],
            light_directions=so["illumination_directions"], cam_colours=so["hdr_illumination_colours"],
            cam_of_ray=so["cam_of_ray"], visibility=visibility, background_illumination=so["hdr_background_colours"],
            weights=weights)  # :797-805
        accumulation, p2p_dist = so["accumulation"], so["p2p_dist"]
        depth = p2p_dist / ray_bundle.metadata["directions_norm"]
        normal, albedo = so["normal"], so["albedo_on_white"]  # :812-813 (white background), from the same reduction pass
        outputs: Dict[str, Any] = {
            "rgb": rgb, "albedo": albedo, "accumulation": accumulation, "depth": depth, "p2p_dist": p2p_dist, "normal": normal,
            "weights": weights, "hdr_background_colours": so["hdr_background_colours"],
            "directions_norm": ray_bundle.metadata["directions_norm"], "sdf_at_termination": sdf_at_termination,
        }
        if self.training:
            outputs["eik_grad"] = fo[FieldHeadNames.GRADIENT]  # :903-904
            outputs.update(so)
        outputs["normal_vis"] = (outputs["normal"] + 1.0) / 2.0
        if "grid_density" in so:
            outputs["grid_density"] = so["grid_density"]
        if "visibility_dict" in so:
            outputs["visibility_batch"] = so["visibility_dict"]["visibility_batch"]
        return outputs

    def get_loss_dict(self, outputs: Dict[str, Any], batch: Dict[str, Any], metrics_dict=None) -> Dict[str, torch.Tensor]:
        """neusky_model.py:933-1062, both branches, through ONE fused kernel each way (ops.MainLossesFn): the train branch's eight
        closed-form terms + the interlevel kernel, or - while the eval latents are fitted and in eval mode - the sky-masked rgb L1
        and the sky-pixel term (:1036-1059).  The two colour-loss options the `neusky` method does not use (rgb_l2_loss,
        cosine_colour_loss, neusky_config.py:104-105) are not built."""
        li = self.config.loss_inclusions
        if li["rgb_l2_loss"] or li["cosine_colour_loss"]:
            raise NotImplementedError("rgb_l2_loss / cosine_colour_loss are outside the `neusky` method (neusky_config.py:104-105)")
        dev = self.device
        return self._fused_loss_dict(outputs, batch["image"].to(dev), batch["mask"].to(dev),
                                     train_branch=self.training and not self.fitting_eval_latents)

    _FUSED_TERMS = ("rgb_l1_loss", "eikonal_loss", "fg_mask_loss", "hashgrid_density_loss", "ground_plane_loss", "sky_pixel_loss",
                    "visibility_sigmoid_loss", "sdf_level_set_visibility_loss")

    def _fused_loss_dict(self, outputs: Dict[str, Any], image: torch.Tensor, mask: torch.Tensor, train_branch: bool = True) -> Dict[str, torch.Tensor]:
        """same formulas, same keys, same scale_dict semantics as the reference; a term whose input is absent stays out of the dict"""
        li = self.config.loss_inclusions
        learn = self.visibility_field is not None and self.visibility_threshold_method == "learnable"
        sdf_t = outputs.get("sdf_at_termination") if (train_branch and li["sdf_level_set_visibility_loss"]) else None
        sky_ok = li["sky_pixel_loss"]["enabled"] and (train_branch or self.config.eval_latent_optimise_method != "nerf_osr_envmap")  # :1051
        present = {
            "rgb_l1_loss": li["rgb_l1_loss"], "eikonal_loss": train_branch and li["eikonal loss"], "fg_mask_loss": train_branch and li["fg_mask_loss"],
            "hashgrid_density_loss": train_branch and li["hashgrid_density_loss"]["enabled"], "ground_plane_loss": train_branch and li["ground_plane_loss"],
            "sky_pixel_loss": sky_ok, "visibility_sigmoid_loss": train_branch and learn, "sdf_level_set_visibility_loss": sdf_t is not None,
        }
        w = outputs["weights"]
        terms = ops.MainLossesFn.apply(
            outputs["rgb"] if present["rgb_l1_loss"] else None, image, mask.float(),
            outputs["eik_grad"] if present["eikonal_loss"] else None,
            w.reshape(w.shape[0], w.shape[1]) if present["fg_mask_loss"] else None,
            outputs["normal"] if present["ground_plane_loss"] else None,
            outputs["hdr_background_colours"] if present["sky_pixel_loss"] else None,
            outputs["grid_density"] if present["hashgrid_density_loss"] else None,
            sdf_t, self.visibility_threshold if present["visibility_sigmoid_loss"] else None,
            self.sky_pixel_loss.alpha if li["sky_pixel_loss"]["enabled"] else 0.0, li["visibility_sigmoid_loss"]["target_min_bias"])
        coefs = self.config.loss_coefficients
        key = (tuple(float(coefs.get(k, 1.0)) if present[k] else 0.0 for k in self._FUSED_TERMS), str(terms.device))
        cv = _COEF_VECTORS.get(key)
        if cv is None:
            cv = _COEF_VECTORS[key] = torch.tensor(key[0], dtype=torch.float32).to(terms.device)
        scaled = terms * cv  # nerfstudio scale_dict: keys missing from the coefficient table ('eikonal_loss') stay unscaled
        ld = LossDict({k: scaled[i] for i, k in enumerate(self._FUSED_TERMS) if present[k]})
        ld.parts = [(terms, cv, 1.0)]  # the objective is formed from the unscaled pieces in one launch (losses.total_loss)
        if train_branch and li["interlevel_loss"]:
            c_il = float(coefs.get("interlevel_loss", 1.0))
            wl, sl_ = outputs["weights_list"], outputs["sbins_list"]
            per_ray = interlevel_per_ray(wl, sl_)
            inv_n = c_il / wl[-1].numel()
            ld["interlevel_loss"] = (per_ray[0] if len(per_ray) == 1 else torch.cat(per_ray)).sum() * inv_n  # :987-988
            ld.parts += [(pr, None, inv_n) for pr in per_ray]
        return ld

    def get_metrics_dict(self, outputs, batch) -> Dict[str, Any]:
        """neusky_model.py:1064-1077"""
        image = batch["image"].to(self.device, torch.float32)
        rgb = outputs["rgb"].detach()
        assert rgb.shape == image.shape, (rgb.shape, image.shape)
        # psnr, s_val and 1 / s_val from one launch (hip.train_metrics)
        v = self.field.deviation_network.variance.detach() if self.training else None
        mv = hip.train_metrics(rgb.contiguous(), image.contiguous(), None, 1.0, v)
        m: Dict[str, Any] = {"psnr": mv[0]}
        if self.training:
            m["s_val"], m["inv_s"] = mv[1:2], mv[2:3]
        if self.training:
            if self.config.visibility_threshold == "learnable" and self.visibility_field is not None:
                m["visibility_threshold"] = self.visibility_threshold.detach()
        return m

    def get_image_metrics_and_images(self, outputs: Dict[str, Any], batch: Dict[str, Any]):
        """neusky_model.py:1079-1335: per-image metrics + the image dict for the viewer / logger.  PSNR, SSIM (torchmetrics'
        structural_similarity_index_measure restated in utils/image_metrics.py) and MSE are computed; LPIPS needs torchmetrics'
        pretrained VGG weights, absent here, and is reported as NaN.  Colour-mapped panels use a plain grey ramp (nerfstudio's
        `colormaps` is not vendored by the reference).  Ground-truth material panels (:1184-1254) need the synthetic EXR layers
        and are produced when the batch carries them."""
        from ..utils.image_metrics import grey_ramp, psnr, ssim
        dev = self.device
        image = batch["image"].to(dev)
        rgb = outputs["rgb"]
        acc = grey_ramp(outputs["accumulation"])
        gt_acc = grey_ramp(batch["mask"][..., 1:2].float().to(dev))  # fg mask (:1088)
        normal = (outputs["normal"] + 1.0) / 2.0
        depth = grey_ramp(outputs["depth"] / outputs["depth"].max().clamp_min(1e-8))
        squared_error = (rgb - image) ** 2
        normalised_error = (squared_error - squared_error.min()) / (squared_error.max() - squared_error.min()).clamp_min(1e-12)
        images_dict = {
            "img": torch.cat([image, rgb], dim=1), "accumulation": torch.cat([gt_acc, acc], dim=1), "depth": depth,
            "normal": torch.cat([(batch["normal"].to(dev) + 1.0) / 2.0, normal], dim=1) if "normal" in batch else normal,
            "normalised_error": grey_ramp(normalised_error.mean(dim=-1, keepdim=True)),
        }
        method = getattr(self.config, "eval_latent_optimise_method", "per_image")
        if method in ["nerf_osr_holdout", "nerf_osr_envmap"]:  # metrics inside the provided test mask only (:1135-1140)
            m = batch["mask"][..., 0:1].float().to(dev)
            rgb, image = rgb * m, image * m
        im4, rgb4 = torch.moveaxis(image, -1, 0)[None], torch.moveaxis(rgb, -1, 0)[None]
        metrics_dict = {"psnr": float(psnr(im4, rgb4)), "ssim": float(ssim(im4, rgb4)), "lpips": float("nan"),
                        "mse": float(torch.mean((im4 - rgb4) ** 2))}
        images_dict["albedo"] = outputs["albedo"]
        if "hdr_background_colours" in outputs:
            images_dict["envmap_from_camera"] = linear_to_sRGB(outputs["hdr_background_colours"])
        fg = batch["mask"][..., 1:2].float().to(dev)
        if "gt_albedo" in batch:  # :1184-1208 (NeRFactor convention: per-channel least-squares rescale on the foreground)
            gt_srgb, pred_srgb = linear_to_sRGB(batch["gt_albedo"].to(dev)), linear_to_sRGB(outputs["albedo"]).clone()
            sel = fg[..., 0] > 0.5
            for ch in range(3):
                g_, p_ = gt_srgb[..., ch][sel], pred_srgb[..., ch][sel]
                if p_.sum() > 1e-8:
                    pred_srgb[..., ch] *= (g_ * p_).sum() / (p_ * p_).sum().clamp(min=1e-8)
            g4, p4 = torch.moveaxis(gt_srgb * fg, -1, 0)[None], torch.moveaxis(pred_srgb * fg, -1, 0)[None]
            metrics_dict["albedo_psnr"], metrics_dict["albedo_ssim"] = float(psnr(g4, p4)), float(ssim(g4, p4))
            images_dict["gt_vs_pred_albedo"] = torch.cat([gt_srgb, pred_srgb], dim=1)
        if "gt_normal" in batch:  # :1210-1236
            gt_n = batch["gt_normal"].to(dev)
            if self.eval_metadata is not None and "orientation_rotation" in self.eval_metadata:
                Rm = self.eval_metadata["orientation_rotation"].to(dev).float()
                gt_n = (Rm @ gt_n.reshape(-1, 3).T).T.reshape(gt_n.shape)
            gn, pn = F.normalize(gt_n, dim=-1), F.normalize(outputs["normal"], dim=-1)
            sel = fg[..., 0] > 0.5
            if sel.any():
                cos = (gn[sel] * pn[sel]).sum(dim=-1).clamp(-1.0, 1.0)
                metrics_dict["normal_mae"] = float((torch.acos(cos) * (180.0 / math.pi)).mean())
            images_dict["gt_vs_pred_normal"] = torch.cat([(gn + 1.0) / 2.0, (pn + 1.0) / 2.0], dim=1)
        return metrics_dict, images_dict

    def generate_ddf_ground_truth(self, ray_bundle: RayBundle, mask_threshold: float = 0.5, log_depth: bool = False,
                                  randoms=None) -> Dict[str, Any]:
        """neusky_model.py:1337-1367: second sampler + field pass on the DDF-fit rays"""
        ray_bundle = self.collider(ray_bundle)
        sub = None if randoms is None or "ddf_jitters" not in randoms else {"jitters": randoms["ddf_jitters"]}
        ray_samples, _, _, _, _ = self._sample(ray_bundle, sub)
        fo = self.field(ray_samples, return_alphas=True, want_albedo=False)  # depth / mask / normals only (:1337-1367)
        weights = fo["weights"]
        p2p, accumulations, normals, _ = self.ray_reductions(
            weights, ray_samples, fo[FieldHeadNames.NORMALS], None,
            max_clamp=2 * self.visibility_field.ddf_radius if self.visibility_field is not None else 0.0)
        mask = (accumulations > mask_threshold).float()
        if log_depth:  # :1354-1355
            p2p = torch.log(p2p + 1e-6)
        return {"ray_bundle": ray_bundle, "accumulations": accumulations, "mask": mask, "termination_dist": p2p, "normals": normals}

    def _eval_fit_bundle(self, datamanager):
        """:1544-1575: the rays of one fitting step and, for nerf_osr_envmap, the per-session z-rotations"""
        method = self.config.eval_latent_optimise_method
        if method == "per_image":
            rb, batch = datamanager.get_eval_image_half_bundle(sample_region=self.config.eval_latent_sample_region)
            return rb, batch, None
        if method == "nerf_osr_holdout":
            rb, batch = datamanager.get_nerfosr_lighting_eval_bundle("compare" if self.config.optimise_compare_eval_scale else "optimise")
            return rb, batch, None
        if method == "nerf_osr_envmap":
            rb, batch = datamanager.get_nerfosr_lighting_eval_bundle("compare")
            gamma = torch.sigmoid(self.eval_rotation) * 2 * math.pi
            cg, sg = torch.cos(gamma), torch.sin(gamma)
            rot = torch.zeros(gamma.shape[0], 3, 3, dtype=gamma.dtype, device=gamma.device)
            rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1], rot[:, 2, 2] = cg, -sg, sg, cg, 1.0
            return rb, batch, rot
        raise NotImplementedError(method)

    def fit_latent_codes_for_eval(self, datamanager, global_step: int, steps: Optional[int] = None, lr: Optional[float] = None,
                                  lr_final: Optional[float] = None, eps: Optional[float] = None, log_every: int = 0,
                                  bundles=None, randoms_per_step=None, use_graph: Optional[bool] = None):
        """neusky_model.py:1503-1588: optimise the evaluation illumination (per-image latents + scale; scale only with
        `optimise_compare_eval_scale`; scale + z-rotation for nerf_osr_envmap) with every other parameter held fixed: Adam,
        exponential decay lr -> lr_final over `steps` iterations (neusky_config.py:142-147: 1e-1 -> 1e-7, 250 steps, eps 1e-15).
        Loss = the eval branch of get_loss_dict (:1036-1059).  Returns the loss trace (device scalars, every `log_every` steps).

        One iteration (zero the gradients, forward, loss, backward) is captured in a HIP graph and replayed: the step's inputs
        are copied into static buffers, the random draws happen on the device inside the graph, and only the Adam update (whose
        learning rate changes every step) is launched from the host.  `bundles` / `randoms_per_step` (tests) inject the ray
        bundles and random draws of every step; injected randoms run eagerly."""
        from ..engine import ExponentialDecaySchedulerConfig
        oc = self.config.eval_latent_optimizer
        steps = int(oc["max_steps"]) if steps is None else steps
        lr = float(oc["lr"]) if lr is None else lr
        lr_final = float(oc["lr_final"]) if lr_final is None else lr_final
        eps = float(oc["eps"]) if eps is None else eps
        method = self.config.eval_latent_optimise_method
        self.fitting_eval_latents = True  # forward() now reads the eval latents (:1506-1507)
        if method == "nerf_osr_envmap":  # :1509-1518
            params = [self.eval_scale, self.eval_rotation]
        elif self.config.optimise_compare_eval_scale:
            params = [self.eval_scale]
        else:
            params = [self.eval_illumination_latents, self.eval_scale]
        with torch.no_grad():  # :1536-1540
            if method != "nerf_osr_envmap":
                self.eval_illumination_latents.zero_()
            self.eval_scale.fill_(1.0)
        for p in params:
            p.requires_grad_(True)
        frozen = [p for p in self.parameters() if p.requires_grad and all(p is not q for q in params)]
        for p in frozen:
            p.requires_grad_(False)
        old_grads = [p.grad for p in params]
        for p in params:
            p.grad = torch.zeros_like(p)
        state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
        sched = ExponentialDecaySchedulerConfig(lr_final=lr_final, max_steps=steps, lr_init=lr)
        trace = []

        def next_bundle(it):
            if bundles is not None:
                rb, batch = bundles[it % len(bundles)]
                return rb, batch, None
            return self._eval_fit_bundle(datamanager)

        def iteration(rb, batch, rot, randoms):
            for p in params:
                p.grad.zero_()
            self.begin_step()
            outputs = self.forward(ray_bundle=rb, step=global_step, rotation=rot, randoms=randoms)
            loss = total_loss(self.get_loss_dict(outputs, batch))
            loss.backward()
            return loss.detach()

        try:
            rb0, batch0, rot0 = next_bundle(0)
            if use_graph is None:
                use_graph = randoms_per_step is None and method != "nerf_osr_envmap"
            graph = None
            if use_graph:
                c = lambda t: t.detach().clone()  # noqa: E731
                srb = RayBundle(origins=c(rb0.origins), directions=c(rb0.directions), pixel_area=c(rb0.pixel_area),
                                camera_indices=c(rb0.camera_indices), metadata={k: c(v) for k, v in rb0.metadata.items()})
                sbatch = {"image": c(batch0["image"]), "mask": c(batch0["mask"])}
                srnd = None
                if randoms_per_step is not None:  # injected draws live in static buffers the graph reads
                    keys = ("jitters", "light_rotation", "grid_perturb", "grid_dirs")
                    srnd = {k: ([c(t) for t in v] if isinstance(v, (list, tuple)) else c(v)) for k, v in randoms_per_step[0].items() if k in keys}
                side = ops.role_stream("capture", self.device)  # warm-up and capture on the package's capture stream (ops.role_stream)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):  # eager warm-up (allocator pools, per-step caches); no parameter is updated by it
                    iteration(srb, sbatch, None, srnd)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(graph, stream=side, capture_error_mode=ops.CAPTURE_MODE):
                        gloss = iteration(srb, sbatch, None, srnd)
                except RuntimeError as exc:  # a host-dependent op inside the iteration: run the fit with host launches instead
                    import warnings
                    warnings.warn(f"fit_latent_codes_for_eval: HIP graph capture refused ({str(exc)[:100]}); running eagerly")
                    torch.cuda.synchronize()
                    graph = None
            for it in range(steps):
                rb, batch, rot = (rb0, batch0, rot0) if it == 0 else next_bundle(it)
                if graph is not None:
                    srb.origins.copy_(rb.origins, non_blocking=True); srb.directions.copy_(rb.directions, non_blocking=True)
                    srb.camera_indices.copy_(rb.camera_indices, non_blocking=True)
                    for k, v in rb.metadata.items():
                        srb.metadata[k].copy_(v, non_blocking=True)
                    sbatch["image"].copy_(batch["image"], non_blocking=True); sbatch["mask"].copy_(batch["mask"], non_blocking=True)
                    if srnd is not None:
                        for k, v in srnd.items():
                            src = randoms_per_step[it][k]
                            for dst_t, src_t in (zip(v, src) if isinstance(v, list) else ((v, src),)):
                                dst_t.copy_(src_t, non_blocking=True)
                    graph.replay()
                    loss = gloss
                else:
                    loss = iteration(rb, batch, rot, None if randoms_per_step is None else randoms_per_step[it])
                for p, (m_, v_) in zip(params, state):
                    hip.adam_step(p.data, p.grad, m_, v_, lr * sched.factor(it), 0.9, 0.999, eps, it + 1)
                if log_every and it % log_every == 0:
                    trace.append(loss.clone())
        finally:
            for p in frozen:
                p.requires_grad_(True)
            for p, g in zip(params, old_grads):
                p.grad = g
            self.fitting_eval_latents = False  # :1588
            ops.retire_graph(locals().get("graph"))  # never destroyed next to its last replay (ops.retire_graph: a runtime use-after-free)
        return trace

    def begin_frame(self, camera_index: int, rotation: Optional[torch.Tensor] = None) -> None:
        """decode the illumination of ONE camera for a whole frame (the reference re-decodes it in each of the
        8100 chunks of a 1080p frame, neusky_model.py:1413-1432; the result is the same)"""
        latents, scales = self.get_illumination_field()
        fixed = self.config.fix_test_illumination_directions
        dirs, sel = self.illumination_sampler.on_device(self.device, apply_random_rotation=False if fixed else None)
        cam = int(camera_index)
        D = dirs.shape[0]
        if rotation is None:
            cols = self.illumination_field.forward_grid(dirs, latents[cam][None], scales[cam][None])
        else:
            cols = self.illumination_field.forward_camera(dirs, latents[cam], scales[cam], rotation)[None]
        # static per-model buffers: a chunk graph captured for one frame stays valid for the next (animation frames
        # only change the camera / rotation, render_animation.py:196-207)
        st = getattr(self, "_frame_static", None)
        if st is None or st[0].shape != dirs.shape or st[0].device != dirs.device:
            st = (torch.empty_like(dirs), torch.empty_like(cols), torch.empty_like(sel))
            self._frame_static = st
            for old in getattr(self, "_chunk_runners", {}).values():
                ops.retire_graph(old.graph)
            self._chunk_runners = {}
        st[0].copy_(dirs); st[1].copy_(cols); st[2].copy_(sel)
        self._frame_illumination = (st[0], st[1], st[2], cam, rotation)
        self._frame_key = (cam, None if rotation is None else tuple(rotation.reshape(-1).tolist()))

    def end_frame(self) -> None:
        self._frame_illumination = None

    @torch.no_grad()
    def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle: RayBundle, show_progress=False, rotation=None, to_cpu=False,
                                          step=None, camera_index: Optional[int] = None, chunk: Optional[int] = None,
                                          use_graph: bool = True) -> Dict[str, torch.Tensor]:
        """neusky_model.py:1369-1501: chunked full-frame render.  The reference chunks at eval_num_rays_per_chunk = 256
        (8100 python iterations per 1080p frame); any chunk size gives the same image, so a larger static chunk is used
        and its forward is captured once in a HIP graph and replayed per chunk (BASELINE config 5)."""
        assert not self.training, "call model.eval() first"
        chunk = chunk or max(self.config.eval_num_rays_per_chunk, 4096)
        shape = camera_ray_bundle.origins.shape[:-1]
        flat = camera_ray_bundle.slice(0, 1 << 62)
        num_rays = flat.origins.shape[0]
        if camera_index is None:
            camera_index = int(flat.camera_indices.reshape(-1)[0]) if flat.camera_indices is not None else 0
        self.begin_frame(camera_index, rotation)
        keys = ["rgb", "albedo", "accumulation", "depth", "p2p_dist", "normal"]
        out = {k: [] for k in keys}
        try:
            # the background term depends on (camera, rotation) through python values baked into a capture, so graphs
            # are cached per (chunk, camera, rotation); eager runners are free to share
            key = (chunk, use_graph, self._frame_key if use_graph else None)
            runner = self._chunk_runners.get(key)
            if runner is None:
                runner = _ChunkRunner(self, chunk, flat, use_graph)
                if len(self._chunk_runners) >= 4:
                    for old in self._chunk_runners.values():
                        ops.retire_graph(old.graph)
                    self._chunk_runners.clear()
                self._chunk_runners[key] = runner
            for i in range(0, num_rays, chunk):
                res = runner.run(flat, i, min(i + chunk, num_rays))
                for k in keys:
                    out[k].append(res[k].cpu() if to_cpu else res[k])
        finally:
            self.end_frame()
        return {k: torch.cat(v).view(*shape, -1) for k, v in out.items()}


class _ChunkRunner:
    """static-shape forward of one render chunk, optionally captured in a HIP graph and replayed"""

    def __init__(self, model: "NeuSkyFactoModel", chunk: int, flat: RayBundle, use_graph: bool):
        self.model, self.chunk, self.graph = model, chunk, None
        dev = flat.origins.device
        self.rb = RayBundle(origins=torch.zeros(chunk, 3, device=dev), directions=torch.zeros(chunk, 3, device=dev),
                            pixel_area=torch.ones(chunk, 1, device=dev), camera_indices=torch.zeros(chunk, 1, dtype=torch.long, device=dev),
                            metadata={"directions_norm": torch.ones(chunk, 1, device=dev)})
        self.rb.directions[:, 2] = 1.0
        if use_graph:
            self._load(flat, 0, min(chunk, flat.origins.shape[0]))
            side = ops.role_stream("capture", dev)  # warm-up and capture on the package's capture stream (ops.role_stream)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.out = model.forward(self.rb)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side, capture_error_mode=ops.CAPTURE_MODE):
                self.out = model.forward(self.rb)

    def __del__(self):
        try:  # (dropped with its model, possibly right behind its last replay: retired, not destroyed here -- ops.retire_graph)
            ops.retire_graph(self.__dict__.pop("graph", None))
        except Exception:  # noqa: BLE001
            pass

    def _load(self, flat: RayBundle, a: int, b: int) -> None:
        n = b - a
        self.rb.origins[:n].copy_(flat.origins[a:b])
        self.rb.directions[:n].copy_(flat.directions[a:b])
        if "directions_norm" in flat.metadata:
            self.rb.metadata["directions_norm"][:n].copy_(flat.metadata["directions_norm"][a:b])
        if n < self.chunk:  # pad the last chunk with copies of its first ray (results discarded)
            self.rb.origins[n:].copy_(self.rb.origins[:1].expand(self.chunk - n, 3))
            self.rb.directions[n:].copy_(self.rb.directions[:1].expand(self.chunk - n, 3))

    def run(self, flat: RayBundle, a: int, b: int) -> Dict[str, torch.Tensor]:
        self._load(flat, a, b)
        if self.graph is not None:
            self.graph.replay()
            out = self.out
        else:
            out = self.model.forward(self.rb)
        return {k: v[:b - a].clone() for k, v in out.items() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == self.chunk}
