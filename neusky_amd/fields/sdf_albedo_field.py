"""SDF + albedo field on the MI355X kernels.

Mirrors `neusky.fields.sdf_albedo_field.SDFAlbedoField` (neusky/fields/sdf_albedo_field.py:80-282):
same constructor arguments, same `forward / get_outputs / get_sdf_at_pos / get_alpha /
deviation_network.get_variance` surface, same parameter names (`encoding.params`, `glin{l}.weight_g/
weight_v/bias`, `clin{l}.*`, `deviation_network.variance`) so a reference state_dict maps 1:1.

What differs underneath: the tcnn hash grid, the weight-normed Softplus geo MLP (inherited from
nerfstudio `SDFField`), `torch.autograd.grad(sdf, x, create_graph=True)` and the colour MLP are one
chain of HIP kernels (`ops.HashEncodeFn` -> `ops.SDFAlbedoFn`) that carries the input Jacobian in
forward mode, so normals and the eikonal term need no autograd double backward.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Type

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from .. import hip, ops
from ..cameras.rays import RaySamples
from ..encoding import HashGridGeometry
from ..field_components.neusky_fieldheadnames import FieldHeadNames, NeuSkyFieldHeadNames
from ..plugin import ConfigBase, FieldBase


class HashEncoding(nn.Module):
    """parameter holder with tcnn's `params` layout (level-major [entry][2] flattened, U(-1e-4, 1e-4) init)."""

    def __init__(self, geom: HashGridGeometry):
        super().__init__()
        self.geom = geom
        self.params = nn.Parameter((torch.rand(geom.n_params * 2) * 2 - 1) * 1e-4)
        self.n_output_dims = geom.out_dim

    @property
    def table(self) -> torch.Tensor:
        return self.params.view(self.geom.n_params, 2)


class WeightNormLinear(nn.Module):
    """nn.utils.weight_norm(nn.Linear) parameterisation (weight_g [out,1], weight_v [out,in], bias)."""

    def __init__(self, in_dim: int, out_dim: int, weight: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None):
        super().__init__()
        lin = nn.Linear(in_dim, out_dim)
        w = lin.weight.data if weight is None else weight
        b = lin.bias.data if bias is None else bias
        self.weight_v = nn.Parameter(w.clone())
        self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True).clone())
        self.bias = nn.Parameter(b.clone())

    def weight(self) -> torch.Tensor:
        return self.weight_g * self.weight_v / self.weight_v.norm(dim=1, keepdim=True)


class LearnedVariance(nn.Module):
    """nerfstudio LearnedVariance (NeuS s-density); used at sdf_albedo_field.py:145, neusky_model.py:1071."""

    def __init__(self, init_val: float):
        super().__init__()
        self.variance = nn.Parameter(init_val * torch.ones(1))

    def get_variance(self) -> torch.Tensor:
        return torch.exp(self.variance * 10.0).clip(1e-6, 1e6)


@dataclass
class SDFAlbedoFieldConfig(ConfigBase):
    """neusky/fields/sdf_albedo_field.py:71-77 + the inherited nerfstudio SDFFieldConfig members the
    `neusky` method sets (neusky/configs/neusky_config.py:66-77)."""

    _target: Type = field(default_factory=lambda: SDFAlbedoField)
    num_layers: int = 2
    hidden_dim: int = 256
    geo_feat_dim: int = 256
    num_layers_color: int = 2
    hidden_dim_color: int = 256
    appearance_embedding_dim: int = 32
    use_appearance_embedding: bool = False
    bias: float = 0.1
    geometric_init: bool = True
    inside_outside: bool = False
    weight_norm: bool = True
    use_grid_feature: bool = True
    divide_factor: float = 2.0
    beta_init: float = 0.1
    encoding_type: str = "hash"
    num_levels: int = 16
    max_res: int = 2048
    base_res: int = 16
    log2_hashmap_size: int = 19
    features_per_level: int = 2
    use_hash: bool = True
    smoothstep: bool = True
    predict_shininess: bool = False
    scene_contraction_order: str = "Linf"
    """contraction the field really receives (SURVEY.md Appendix A.3: the parent model hands over L-inf)."""

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class SDFAlbedoField(FieldBase):
    config: SDFAlbedoFieldConfig

    def __init__(self, config: SDFAlbedoFieldConfig, aabb: torch.Tensor, num_images: int,
                 use_average_appearance_embedding: bool = False, spatial_distortion=None) -> None:
        nn.Module.__init__(self)  # not the nerfstudio base's constructor (neusky_amd/plugin.py)
        c = self.config = config
        if c.predict_shininess:
            raise NotImplementedError("predict_shininess (Blinn-Phong) is outside the neusky config (neusky_config.py:76)")
        if not (c.num_layers == 2 and c.num_layers_color == 2 and c.encoding_type == "hash" and c.use_hash and
                c.weight_norm and c.use_grid_feature and c.features_per_level == 2):
            raise NotImplementedError("the HIP field is specialised for the `neusky` method shapes (neusky_config.py:66-77)")
        self.aabb = nn.Parameter(aabb.clone(), requires_grad=False)
        self.spatial_distortion = spatial_distortion
        self.num_images = num_images
        self.embedding_appearance = nn.Embedding(num_images, c.appearance_embedding_dim)  # sdf_albedo_field.py:110 (unused)
        self.embedding_appearance.weight.requires_grad_(False)
        self.use_grid_feature = c.use_grid_feature
        self.divide_factor = c.divide_factor
        self.geom = HashGridGeometry(n_levels=c.num_levels, log2_hashmap_size=c.log2_hashmap_size, base_res=c.base_res,
                                     max_res=c.max_res, smoothstep=c.smoothstep)
        self.encoding = HashEncoding(self.geom)  # sdf_albedo_field.py:119-130
        self.grid_mode = hip.MODE_CONTRACT_L2 if c.scene_contraction_order == "L2" else hip.MODE_CONTRACT_LINF

        # ---- geometric network (nerfstudio SDFField.initialize_geo_layers)
        pe_dim = 36
        dims = [3 + pe_dim + self.geom.out_dim] + [c.hidden_dim] * c.num_layers + [1 + c.geo_feat_dim]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1]
            w = torch.empty(out_dim, dims[l])
            b = torch.zeros(out_dim)
            if c.geometric_init:
                if l == self.num_layers - 2:
                    sign = -1.0 if c.inside_outside else 1.0
                    nn.init.normal_(w, mean=sign * math.sqrt(math.pi) / math.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(b, -sign * c.bias)
                elif l == 0:
                    nn.init.constant_(w, 0.0)
                    nn.init.normal_(w[:, :3], 0.0, math.sqrt(2) / math.sqrt(out_dim))
                else:
                    nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
            else:
                lin = nn.Linear(dims[l], out_dim)
                w, b = lin.weight.data, lin.bias.data
            setattr(self, f"glin{l}", WeightNormLinear(dims[l], out_dim, w, b))
        self.deviation_network = LearnedVariance(init_val=c.beta_init)  # sdf_albedo_field.py:145

        # ---- colour network (sdf_albedo_field.py:147-161)
        cdims = [3 + pe_dim + c.geo_feat_dim] + [c.hidden_dim_color] * c.num_layers_color + [3]
        self.num_layers_color = len(cdims)
        for l in range(self.num_layers_color - 1):
            setattr(self, f"clin{l}", WeightNormLinear(cdims[l], cdims[l + 1]))
        self._cos_anneal_ratio = 1.0  # sdf_albedo_field.py:167
        self.softplus_beta = 100.0  # :163

    # ------------------------------------------------------------------ weight preparation (tiny torch ops)
    def invalidate_weight_cache(self) -> None:
        """call once per optimisation step (before the first use): the weight-normed / permuted / padded matrices are
        prepared once and shared by every pass of the step (main rays, DDF-fit rays, grid probe, sdf probes)"""
        self._wcache = {}

    def _geo_weights(self):
        c = getattr(self, "_wcache", None)
        if c is not None and "geo" in c and torch.is_grad_enabled() == c["geo_grad"]:
            return c["geo"]
        w = self._geo_weights_uncached()
        if c is not None:
            c["geo"], c["geo_grad"] = w, torch.is_grad_enabled()
        return w

    def _colour_weights(self):
        c = getattr(self, "_wcache", None)
        if c is not None and "col" in c and torch.is_grad_enabled() == c["col_grad"]:
            return c["col"]
        w = self._colour_weights_uncached()
        if c is not None:
            c["col"], c["col_grad"] = w, torch.is_grad_enabled()
        return w

    def _wn(self, lin: "WeightNormLinear", key: str, rows, cols) -> torch.Tensor:
        """weight-normed matrix of one layer in the layout its consumer wants: one launch (ops.WeightNormFn)"""
        dev = lin.weight_v.device
        maps = self._wn_maps.get((key, str(dev))) if hasattr(self, "_wn_maps") else None
        if maps is None:
            if not hasattr(self, "_wn_maps"):
                self._wn_maps = {}
            o, i = lin.weight_v.shape
            maps = self._wn_maps[(key, str(dev))] = ops.weight_norm_maps(o, i, rows(o, i), cols(o, i), dev)
        w = ops.WeightNormFn.apply(lin.weight_v, lin.weight_g, *maps)
        w._nsky_src = ("wn", id(lin.weight_v), key)  # (the packed weight streams keep one persistent buffer per source parameter)
        return w

    def _geo_weights_uncached(self):
        ident = lambda n, m: list(range(n))  # noqa: E731
        pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
        W0 = self._wn(self.glin0, "g0", ident, lambda o, i: list(range(i)) + [-1] * (pad4(i) - i))
        W1 = self._wn(self.glin1, "g1", ident, lambda o, i: list(range(i)))
        # rows: [feat | sdf | 0 0 0]
        W2p = self._wn(self.glin2, "g2", lambda o, i: list(range(1, o)) + [0, -1, -1, -1], lambda o, i: list(range(i)))
        b2 = self.glin2.bias
        b2p = torch.cat([b2[1:], b2[:1], b2.new_zeros(3)], 0).contiguous()
        return (W0, self.glin0.bias.contiguous(), W1, self.glin1.bias.contiguous(), W2p, b2p)

    def _colour_weights_uncached(self):
        ident = lambda n, m: list(range(n))  # noqa: E731
        pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
        # columns [x(3) PE(36) feat(GF)] -> [feat | 0 0 0 0 | x PE | 0]
        Wc0p = self._wn(self.clin0, "c0", ident, lambda o, i: list(range(39, i)) + [-1] * 4 + list(range(39)) + [-1])
        Wc1 = self._wn(self.clin1, "c1", ident, lambda o, i: list(range(i)))
        Wc2p = self._wn(self.clin2, "c2", lambda o, i: list(range(o)) + [-1] * (pad4(o) - o), lambda o, i: list(range(i)))
        return (Wc0p, self.clin0.bias.contiguous(), Wc1, self.clin1.bias.contiguous(), Wc2p, ops.pad_bias(self.clin2.bias))

    def _encode(self, positions_flat: torch.Tensor, tangents: bool, need_dx: bool) -> torch.Tensor:
        return ops.HashEncodeFn.apply(positions_flat, self.encoding.table, self.geom, self.grid_mode, True, 6, 5.0,
                                      tangents, need_dx)

    # ------------------------------------------------------------------ reference surface
    def set_cos_anneal_ratio(self, anneal: float) -> None:
        self._cos_anneal_ratio = anneal

    def get_sdf_at_pos(self, positions: torch.Tensor) -> torch.Tensor:
        """sdf_albedo_field.py:169-174 -> [P,1]; differentiable w.r.t. the field AND the positions."""
        x = positions.reshape(-1, 3)
        E = self._encode(x, False, x.requires_grad)
        sdf = ops.SDFValueFn.apply(E, *self._geo_weights(), self.softplus_beta, True)
        return sdf[:, None]

    def field_values(self, positions_flat: torch.Tensor, want_albedo: bool = True):
        """sdf [N], gradients [N,3], albedo [N,3] at flat positions (sdf_albedo_field.py:225-246).  want_albedo=False: the
        colour net is skipped each way and albedo comes back as zeros (geometry-only passes: DDF-fit ground truth, grid probe)."""
        ET = self._encode(positions_flat.detach(), True, False)
        return ops.field_apply(ET, *self._geo_weights(), *self._colour_weights(), self.softplus_beta, want_albedo)

    def get_colors(self, points: torch.Tensor, geo_features: torch.Tensor) -> torch.Tensor:
        """sdf_albedo_field.py:185-209: albedo of the colour network at `points` given their geometric features:
        [x | PE6(x) | feat] -> Linear+ReLU -> Linear+ReLU... -> sigmoid.  (The training step reaches the same three dense
        layers fused behind the geo network inside ops.SDFAlbedoFn; this entry point serves callers written against the reference.)"""
        from .directional_distance_field import nerf_encoding
        x = points.reshape(-1, 3)
        feat = geo_features.reshape(x.shape[0], -1)
        Wc0p, bc0, Wc1, bc1, Wc2p, bc2 = self._colour_weights()
        GF = feat.shape[1]
        cin = x.new_zeros(x.shape[0], Wc0p.shape[1])  # columns [feat | 0 0 0 0 | x PE | 0] (see _colour_weights_uncached)
        cin[:, :GF] = feat
        cin[:, GF + 4:GF + 4 + 39] = torch.cat([x, nerf_encoding(x, 6, 5.0)], -1)
        h0 = ops.DenseFn.apply(cin, Wc0p, bc0, Wc0p.shape[0], "relu", True)
        h1 = ops.DenseFn.apply(h0, Wc1, bc1, Wc1.shape[0], "relu", True)
        out = ops.DenseFn.apply(h1, Wc2p, bc2, 3, "none", True)
        return torch.sigmoid(out[:, :3]).reshape(*points.shape[:-1], 3)

    def get_alpha(self, ray_samples: RaySamples, sdf: Optional[torch.Tensor] = None, gradients: Optional[torch.Tensor] = None):
        """nerfstudio SDFField.get_alpha for isolated samples (used by the hash-grid density probe,
        neusky_model.py:732): one sample per 'ray' -> alpha [P,1]."""
        x = ray_samples.frustums.get_start_positions().reshape(-1, 3)
        d = ray_samples.frustums.directions.reshape(-1, 3)
        if sdf is None or gradients is None:
            sdf, gradients, _ = self.field_values(x)
        P = x.shape[0]
        deltas = ray_samples.deltas.reshape(-1).expand(P) if ray_samples.deltas.numel() != P else ray_samples.deltas.reshape(-1)
        starts = torch.zeros(P, 1, device=x.device)
        w, _, _, _ = ops.NeusWeightsFn.apply(sdf.reshape(P, 1), gradients.reshape(P, 1, 3), d, starts,
                                             deltas.reshape(P, 1).contiguous(), self.deviation_network.variance,
                                             self._cos_anneal_ratio)
        return w  # with a single sample, weight == alpha

    def get_outputs(self, ray_samples: RaySamples, density_embedding=None, return_alphas: bool = False,
                    want_albedo: bool = True, extra_points: Optional[torch.Tensor] = None) -> Dict:
        """sdf_albedo_field.py:211-269.  Besides the reference keys, `weights` [R,S,1], `bg_transmittance` [R,1],
        `accumulation` [R,1] and `p2p_dist` [R,1] come out of the same fused NeuS kernel when return_alphas.
        extra_points [P,3]: isolated points evaluated in the same pass; their sdf / gradients come back as `extra_sdf` [P] and
        `extra_gradients` [P,3]."""
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        fr = ray_samples.frustums
        R, S = fr.origins.shape[:2]
        x = fr.get_start_positions().reshape(-1, 3)
        n_main = x.shape[0]
        if extra_points is not None:  # isolated points evaluated in the same pass (tail rows): the model's hash-grid density probe
            x = torch.cat([x, extra_points.reshape(-1, 3)], 0)
        sdf, grad, albedo = self.field_values(x, want_albedo)
        extra = None
        if extra_points is not None:
            if sdf.requires_grad:
                sdf, e_sdf = ops.SplitRowsFn.apply(sdf, n_main)
                grad, e_grad = ops.SplitRowsFn.apply(grad, n_main)
                albedo = ops.SplitRowsFn.apply(albedo, n_main)[0] if albedo.requires_grad else albedo[:n_main]
                extra = (e_sdf, e_grad)
            else:
                extra = (sdf[n_main:], grad[n_main:])
                sdf, grad, albedo = sdf[:n_main], grad[:n_main], albedo[:n_main]
        outputs = {
            NeuSkyFieldHeadNames.ALBEDO: albedo.view(R, S, 3),
            FieldHeadNames.SDF: sdf.view(R, S, 1),
            FieldHeadNames.NORMALS: ops.NormalizeFn.apply(grad).view(R, S, 3),
            FieldHeadNames.GRADIENT: grad.view(R, S, 3),
        }
        if return_alphas:
            ray_dirs = fr.directions[:, 0].contiguous()
            w, tbg, acc, dep = ops.NeusWeightsFn.apply(sdf.view(R, S), grad.view(R, S, 3), ray_dirs,
                                                       fr.starts.reshape(R, S), fr.ends.reshape(R, S),
                                                       self.deviation_network.variance, self._cos_anneal_ratio)
            outputs["weights"] = w[..., None]
            outputs["bg_transmittance"] = tbg[:, None]
            outputs["accumulation"] = acc[:, None]
            outputs["p2p_dist_unclipped"] = dep[:, None]
        if extra is not None:
            outputs["extra_sdf"], outputs["extra_gradients"] = extra
        return outputs

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False, return_alphas: bool = False,
                want_albedo: bool = True, extra_points: Optional[torch.Tensor] = None) -> Dict:
        return self.get_outputs(ray_samples, return_alphas=return_alphas, want_albedo=want_albedo, extra_points=extra_points)
