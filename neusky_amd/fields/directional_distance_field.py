"""Directional distance field (DDF) on the MI355X kernels.

Mirrors `neusky.fields.directional_distance_field.DirectionalDistanceField`
(neusky/fields/directional_distance_field.py:94-315) for the configured branch only
(neusky/configs/neusky_config.py:163-177): hash(position) + NeRF(direction), FiLM conditioning, `ddf` head.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Type

import torch
from torch import nn

from .. import hip, ops
from ..cameras.rays import RaySamples
from ..encoding import HashGridGeometry
from ..field_components.neusky_fieldheadnames import NeuSkyFieldHeadNames
from ..utils.siren import FiLMSiren
from .sdf_albedo_field import HashEncoding
from ..plugin import ConfigBase, FieldBase


def nerf_encoding(x: torch.Tensor, num_freq: int, max_exp: float, include_input: bool = False) -> torch.Tensor:
    """nerfstudio NeRFEncoding with min_freq_exp = 0 (torch ops; only used on small batches)."""
    freqs = 2.0 ** torch.linspace(0.0, max_exp, num_freq, device=x.device)
    xs = (2.0 * math.pi * x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    enc = torch.sin(torch.cat([xs, xs + math.pi / 2.0], -1))
    return torch.cat([x, enc], -1) if include_input else enc


@dataclass
class DirectionalDistanceFieldConfig(ConfigBase):
    """neusky/fields/directional_distance_field.py:47-91"""

    _target: Type = field(default_factory=lambda: DirectionalDistanceField)
    position_encoding_type: str = "hash"
    direction_encoding_type: str = "nerf"
    conditioning: str = "FiLM"
    termination_output_activation: str = "sigmoid"
    probability_of_hit_output_activation: str = "sigmoid"
    hidden_layers: int = 5
    hidden_features: int = 256
    mapping_layers: int = 5
    mapping_features: int = 256
    num_attention_heads: int = 8
    num_attention_layers: int = 6
    out_features: int = 3
    last_layer_linear: bool = True
    first_omega_0: float = 30.0
    hidden_omega_0: float = 30.0
    predict_probability_of_hit: bool = False
    ddf_type: str = "ddf"

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class DirectionalDistanceField(FieldBase):
    config: DirectionalDistanceFieldConfig

    def __init__(self, config: DirectionalDistanceFieldConfig, ddf_radius: float = 1.0) -> None:
        nn.Module.__init__(self)  # not the nerfstudio base's constructor (neusky_amd/plugin.py)
        c = self.config = config
        self.ddf_radius = ddf_radius
        if c.conditioning != "FiLM" or c.ddf_type != "ddf" or c.position_encoding_type != "hash" or \
                c.direction_encoding_type != "nerf" or c.predict_probability_of_hit or c.termination_output_activation != "sigmoid":
            # same exception type the reference raises for unsupported enum values (:178,181,218,255)
            raise NotImplementedError("only the FiLM / hash / nerf / ddf branch selected by neusky_config.py:163-177 is built")
        # directional_distance_field.py:139-156: L=16, F=2, T=2^19, 16->2048, Linear interpolation
        self.geom = HashGridGeometry(n_levels=16, log2_hashmap_size=19, base_res=16, max_res=2048, smoothstep=False)
        self.position_encoding = HashEncoding(self.geom)
        self.ddf = FiLMSiren(in_dim=3 + 12, hidden_layers=c.hidden_layers, hidden_features=c.hidden_features,
                             mapping_network_in_dim=3 + self.geom.out_dim, mapping_network_layers=c.mapping_layers,
                             mapping_network_features=c.mapping_features, out_dim=1, outermost_linear=c.last_layer_linear)

    def direction_rows(self, local_dirs: torch.Tensor) -> torch.Tensor:
        """[d | NeRF2(d)] zero padded to 16 columns (:270-271)"""
        row = torch.cat([local_dirs, nerf_encoding(local_dirs, 2, 2.0)], -1)
        return torch.nn.functional.pad(row, (0, 1)).contiguous()

    def condition_rows(self, sphere_positions: torch.Tensor) -> torch.Tensor:
        """[p | hash(p)] :267-268"""
        return ops.HashEncodeFn.apply(sphere_positions, self.position_encoding.table, self.geom, hip.MODE_RAW, True, 0, 0.0, False, False)

    def forward_encoded(self, xrow: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
        out = self.ddf(xrow, cond, padded_output=True)
        return ops.SigmoidColumnFn.apply(out, 2 * self.ddf_radius)  # :297-299

    def forward_rows(self, sphere_positions: torch.Tensor, xrow: torch.Tensor) -> torch.Tensor:
        """fast path: xrow [M,16] already encoded (nsky_visibility_rays) -> expected termination distance [M]"""
        return self.forward_encoded(xrow, self.condition_rows(sphere_positions))

    def get_outputs(self, ray_samples: RaySamples) -> Dict:
        origins = ray_samples.frustums.origins.reshape(-1, 3).contiguous()
        local_dirs = ray_samples.frustums.directions.reshape(-1, 3)
        t = self.forward_rows(origins, self.direction_rows(local_dirs))
        return {NeuSkyFieldHeadNames.TERMINATION_DISTANCE: t}

    def forward(self, ray_samples: RaySamples) -> Dict:
        return self.get_outputs(ray_samples)
