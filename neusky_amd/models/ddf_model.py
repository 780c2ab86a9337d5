"""DDF model (sky-visibility field) - mirrors neusky/models/ddf_model.py:89-493 for the hot path:
`get_localised_transforms` (:158-181), `get_outputs` (:183-369), `get_loss_dict` (:407-493),
`get_param_groups` (:151-156).  Image / metric code (:550-674) is out of scope."""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Type

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import Parameter

from ..cameras.rays import Frustums, RayBundle, RaySamples
from ..field_components.neusky_fieldheadnames import NeuSkyFieldHeadNames
from ..fields.directional_distance_field import DirectionalDistanceFieldConfig
from .. import ops
from ..model_components.losses import LossDict, scale_dict

_COEF_VECTORS: Dict[tuple, torch.Tensor] = {}
from ..utils.utils import device_rng, ray_sphere_intersection
from ..plugin import ConfigBase, ModelBase


@dataclass
class DDFModelConfig(ConfigBase):
    """neusky/models/ddf_model.py:53-86 with the values of neusky/configs/neusky_config.py:162-206"""

    _target: Type = field(default_factory=lambda: DDFModel)
    ddf_field: DirectionalDistanceFieldConfig = field(default_factory=DirectionalDistanceFieldConfig)
    compute_normals: bool = False
    include_depth_loss_scene_center_weight: bool = True
    scene_center_weight_exp: float = 3.0
    scene_center_weight_include_z: bool = False
    mask_to_circumference: bool = False
    inverse_depth_weight: bool = False
    log_depth: bool = False
    eval_num_rays_per_chunk: int = 1024
    loss_inclusions: Dict[str, bool] = field(default_factory=lambda: {
        "depth_l1_loss": True, "depth_l2_loss": False, "sdf_l1_loss": False, "sdf_l2_loss": True, "prob_hit_loss": False,
        "normal_loss": False, "multi_view_loss": True, "sky_ray_loss": True})
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {
        "depth_l1_loss": 1.0, "depth_l2_loss": 0.0, "sdf_l1_loss": 1.0, "sdf_l2_loss": 0.01, "prob_hit_loss": 0.01,
        "normal_loss": 1.0, "multi_view_loss": 0.01, "sky_ray_loss": 1.0})

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class DDFModel(ModelBase):
    config: DDFModelConfig

    def __init__(self, config: DDFModelConfig, ddf_radius: float, **kwargs) -> None:
        nn.Module.__init__(self)  # not the nerfstudio base's constructor (neusky_amd/plugin.py)
        self.config = config
        self.ddf_radius = ddf_radius
        if config.compute_normals:
            raise NotImplementedError("compute_normals is disabled in the neusky config (neusky_config.py:199)")
        self.field = config.ddf_field.setup(ddf_radius=ddf_radius)

    def get_param_groups(self) -> Dict[str, List[Parameter]]:
        return {"ddf_field": list(self.field.parameters())}

    def get_localised_transforms(self, positions: torch.Tensor) -> torch.Tensor:
        """ddf_model.py:158-181 (torch ops; the [R*Dv] visibility rays use the fused HIP kernel instead)"""
        up = torch.zeros_like(positions)
        up[:, 2] = 1.0
        y = -positions
        x = torch.linalg.cross(up, y, dim=-1)
        xn = x.norm(dim=-1, keepdim=True)
        # a position ON the z axis: 0 / 0 in the reference; any unit vector across the axis completes the frame (as csrc/samplers.hip does)
        x = torch.where(xn > 0, x / xn.clamp_min(1e-38), torch.tensor([1.0, 0.0, 0.0], dtype=x.dtype, device=x.device).expand_as(x))
        z = torch.linalg.cross(y, x, dim=-1)
        z = z / z.norm(dim=-1, keepdim=True)
        return torch.stack((x, y, z), dim=-1)

    def query(self, positions: torch.Tensor, directions: torch.Tensor) -> torch.Tensor:
        """world rays on the sphere -> expected termination distance [M] (:193-219)"""
        rot = self.get_localised_transforms(positions)
        local = torch.einsum("ijl,ij->il", rot, directions)
        rs = RaySamples(frustums=Frustums(origins=positions, directions=local, starts=torch.zeros_like(positions),
                                          ends=torch.zeros_like(positions), pixel_area=torch.ones_like(positions[..., 0])))
        return self.field.forward(rs)[NeuSkyFieldHeadNames.TERMINATION_DISTANCE]

    def prepare_queries(self, ray_bundle: RayBundle, batch, mv_points: Optional[torch.Tensor] = None) -> Dict[str, Any]:
        """The inputs of EVERY DDF evaluation get_outputs will need (rays | multi-view | sky), for ops.DDFQueryRowsFn: the
        pipeline hands them to NeuSkyFactoModel.compute_visibility so that they ride in the same chain launches as the 262 144
        visibility rows instead of a second chain of small launches (and are encoded by one kernel instead of ~100 torch ops)."""
        c = self.config
        positions = ray_bundle.origins.reshape(-1, 3).contiguous()
        directions = ray_bundle.directions.reshape(-1, 3).contiguous()
        want_mv = bool(c.loss_inclusions["multi_view_loss"] and self.training and batch is not None)
        want_sky = bool(c.loss_inclusions["sky_ray_loss"] and self.training and batch is not None)
        q_seed, q_counter = device_rng(self, "ddf_query_rows", 1, positions.device)
        rng = {"counter": q_counter, "seed": q_seed}
        sky = batch["sky_ray_bundle"] if want_sky else None
        return {"positions": positions, "directions": directions, "term_dist": batch["termination_dist"] if batch is not None else None,
                "want_mv": want_mv, "mv_points_in": None if mv_points is None else mv_points.to(positions).contiguous(),
                "seed": rng["seed"], "counter": rng["counter"],
                "sky_o": None if sky is None else sky.origins.reshape(-1, 3).contiguous(),
                "sky_d": None if sky is None else sky.directions.reshape(-1, 3).contiguous(),
                "want_weight": bool(c.include_depth_loss_scene_center_weight and self.training and batch is not None),
                "weight_exp": float(c.scene_center_weight_exp), "weight_include_z": bool(c.scene_center_weight_include_z)}

    def get_outputs(self, ray_bundle: RayBundle, batch, neusky, stop_gradients: bool = True,
                    mv_points: Optional[torch.Tensor] = None,
                    precomputed: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """precomputed = what NeuSkyFactoModel.compute_visibility_compact left of the fit rows it evaluated (prepare_queries):
        {"t_main", "t_mv", "t_sky": DDF outputs of the three spans, "sky_gt", "distance_weight", "mv_points", "sdf_main"}."""
        positions = ray_bundle.origins.reshape(-1, 3)
        directions = ray_bundle.directions.reshape(-1, 3)
        outputs: Dict[str, Any] = {}
        c = self.config
        if precomputed is not None:  # every evaluation already ran with the visibility rows (prepare_queries): assemble
            outputs["expected_termination_dist"] = precomputed["t_main"]
            if c.include_depth_loss_scene_center_weight and self.training and batch is not None:
                outputs["distance_weight"] = precomputed["distance_weight"]
            if (c.loss_inclusions["sdf_l1_loss"] or c.loss_inclusions["sdf_l2_loss"]) and self.training:
                if precomputed.get("sdf_main") is not None:
                    outputs["sdf_at_termination"] = precomputed["sdf_main"]
                elif neusky is not None:
                    term = positions + directions * precomputed["t_main"].unsqueeze(-1)
                    with torch.no_grad():
                        outputs["sdf_at_termination"] = neusky.field.get_sdf_at_pos(term).detach()
            if c.loss_inclusions["multi_view_loss"] and self.training and batch is not None:
                outputs["multi_view_termintation_dist"] = batch["termination_dist"]  # (sic) :321
                outputs["multi_view_expected_termination_dist"] = precomputed["t_mv"]
            if c.loss_inclusions["sky_ray_loss"] and self.training and batch is not None:
                outputs["sky_ray_termination_dist"] = precomputed["sky_gt"]  # :343
                outputs["sky_ray_expected_termination_dist"] = precomputed["t_sky"]
            return outputs
        # The reference issues up to three separate DDF evaluations here (the rays themselves :217, the multi-view
        # rays :319, the sky rays :360).  Their inputs do not depend on each other's outputs, so they are gathered
        # first and evaluated as ONE batch (one chain of GEMM launches instead of three).
        q_pos, q_dir, spans = [positions], [directions], {"main": (0, positions.shape[0])}
        want_mv = c.loss_inclusions["multi_view_loss"] and self.training and batch is not None
        want_sky = c.loss_inclusions["sky_ray_loss"] and self.training and batch is not None
        if want_mv:  # :279-317
            gt_pts = positions + directions * batch["termination_dist"].repeat(1, 3)
            if mv_points is None:
                theta = 2 * torch.pi * torch.rand(gt_pts.shape[0], device=gt_pts.device)  # drawn on the device (no host round trip)
                phi = torch.acos(2 * torch.rand(gt_pts.shape[0], device=gt_pts.device) - 1)
                mv_points = torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], 1)
            pts = mv_points.to(gt_pts).clone()
            pts[:, 2] = torch.abs(pts[:, 2])
            dvec = gt_pts - pts
            dlen = torch.norm(dvec, dim=-1)
            n0 = sum(t.shape[0] for t in q_pos)
            q_pos.append(pts); q_dir.append(dvec / dlen.unsqueeze(-1)); spans["mv"] = (n0, n0 + pts.shape[0])
        if want_sky:  # :324-358
            sky = batch["sky_ray_bundle"]
            o, d = sky.origins.reshape(-1, 3), sky.directions.reshape(-1, 3)
            sp = ray_sphere_intersection(o, d, self.ddf_radius)
            n0 = sum(t.shape[0] for t in q_pos)
            q_pos.append(sp); q_dir.append(-d); spans["sky"] = (n0, n0 + sp.shape[0])
        t_all = self.query(torch.cat(q_pos, 0), torch.cat(q_dir, 0))
        expected = t_all[spans["main"][0]:spans["main"][1]]
        outputs["expected_termination_dist"] = expected
        if c.include_depth_loss_scene_center_weight and self.training and batch is not None:  # :224-238
            dist = positions.norm(dim=-1) if c.scene_center_weight_include_z else positions[..., :2].norm(dim=-1)
            outputs["distance_weight"] = 1.0 - (dist / self.ddf_radius) ** c.scene_center_weight_exp
        if (c.loss_inclusions["sdf_l1_loss"] or c.loss_inclusions["sdf_l2_loss"]) and self.training:  # :241-254
            if neusky is not None:
                term = positions + directions * expected.unsqueeze(-1)
                if stop_gradients:
                    with torch.no_grad():
                        sdf = neusky.field.get_sdf_at_pos(term).detach()
                else:
                    sdf = neusky.field.get_sdf_at_pos(term)
                outputs["sdf_at_termination"] = sdf
            elif batch is not None and "sdf_at_termination" in batch:
                outputs["sdf_at_termination"] = batch["sdf_at_termination"]
        if want_mv:
            outputs["multi_view_termintation_dist"] = batch["termination_dist"]  # (sic) :321
            outputs["multi_view_expected_termination_dist"] = t_all[spans["mv"][0]:spans["mv"][1]]
        if want_sky:
            outputs["sky_ray_termination_dist"] = torch.norm(o - sp, dim=-1)  # :343
            outputs["sky_ray_expected_termination_dist"] = t_all[spans["sky"][0]:spans["sky"][1]]
        return outputs

    def forward(self, ray_bundle: RayBundle, batch, neusky, stop_gradients: bool = True, **kw) -> Dict[str, torch.Tensor]:
        return self.get_outputs(ray_bundle, batch, neusky, stop_gradients=stop_gradients, **kw)

    def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
        """ddf_model.py:381-405 (depth PSNR over [0, ddf_radius])"""
        exp_d = outputs["expected_termination_dist"].detach()
        mask = batch["mask"].to(exp_d.device, torch.float32)
        gt = batch["termination_dist"].to(exp_d.device).detach()
        assert mask.numel() == exp_d.numel() and gt.numel() == exp_d.numel(), (mask.shape, gt.shape, exp_d.shape)
        from .. import hip
        return {"depth_psnr": hip.train_metrics(exp_d.contiguous(), gt.contiguous(), mask.contiguous(), self.ddf_radius**2)[0]}

    def get_loss_dict(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
        """ddf_model.py:407-493"""
        c = self.config
        loss_dict: Dict[str, torch.Tensor] = {}
        li = c.loss_inclusions
        exp_d = outputs["expected_termination_dist"]
        if c.inverse_depth_weight and not c.include_depth_loss_scene_center_weight:
            raise NotImplementedError("inverse_depth_weight without the scene-centre weight (ddf_model.py:434-436) is outside the `neusky` method")
        # one launch each way for the five closed-form terms (ops.DDFLossesFn; same formulas, keys and scaling)
        want_sdf = (li["sdf_l2_loss"] or li["sdf_l1_loss"])
        mv = li["multi_view_loss"]
        sky = li["sky_ray_loss"]
        flags = dict(want_depth=int(li["depth_l1_loss"]), want_sdf_l2=int(li["sdf_l2_loss"]), want_sdf_l1=int(li["sdf_l1_loss"]),
                     mask_to_circumference=int(c.mask_to_circumference), inverse_depth_weight=int(c.inverse_depth_weight),
                     radius=float(self.ddf_radius))
        terms = ops.DDFLossesFn.apply(
            exp_d, batch["termination_dist"], batch["mask"],
            outputs["distance_weight"] if c.include_depth_loss_scene_center_weight else None,
            outputs["sdf_at_termination"] if want_sdf else None,
            outputs["multi_view_expected_termination_dist"] if mv else None, outputs["multi_view_termintation_dist"] if mv else None,
            outputs["sky_ray_expected_termination_dist"] if sky else None, outputs["sky_ray_termination_dist"] if sky else None, flags)
        names = ("depth_l1_loss", "sdf_l2_loss", "sdf_l1_loss", "multi_view_loss", "sky_ray_loss")
        present = (li["depth_l1_loss"], li["sdf_l2_loss"], li["sdf_l1_loss"], mv, sky)
        key = (tuple(float(c.loss_coefficients.get(k, 1.0)) if p else 0.0 for k, p in zip(names, present)), str(terms.device))
        cv = _COEF_VECTORS.get(key)
        if cv is None:
            cv = _COEF_VECTORS[key] = torch.tensor(key[0], dtype=torch.float32).to(terms.device)
        scaled = terms * cv
        out = LossDict({k: scaled[i] for i, (k, p) in enumerate(zip(names, present)) if p})
        out.parts = [(terms, cv, 1.0)]  # (the objective: losses.total_loss, one launch for both models' terms)
        return out
