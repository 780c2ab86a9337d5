"""Minimal RayBundle / RaySamples / Frustums carrying exactly the attributes the NeuSky hot path
touches.  Same field names as nerfstudio.cameras.rays so that code written
against the reference reads the same; when nerfstudio is installed its own classes can be passed in
instead (only attribute access is used)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, Optional

import torch


@dataclass
class Frustums:
    origins: torch.Tensor
    directions: torch.Tensor
    starts: torch.Tensor
    ends: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None

    def get_positions(self) -> torch.Tensor:
        return self.origins + self.directions * (self.starts + self.ends) / 2

    def get_start_positions(self) -> torch.Tensor:
        return self.origins + self.directions * self.starts

    @property
    def shape(self):
        return self.origins.shape[:-1]


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[torch.Tensor] = None
    deltas: Optional[torch.Tensor] = None
    spacing_starts: Optional[torch.Tensor] = None
    spacing_ends: Optional[torch.Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, torch.Tensor]] = None

    @property
    def shape(self):
        return self.frustums.shape

    def get_weights_and_transmittance_from_alphas(self, alphas: torch.Tensor):
        """alphas [R,S,1] -> weights [R,S,1], transmittance [R,S+1,1] (neusky_model.py:565-568)."""
        T = torch.cumprod(torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas + 1e-7], 1), 1)
        return alphas * T[:, :-1], T


@dataclass
class RayBundle:
    origins: torch.Tensor
    directions: torch.Tensor
    pixel_area: Optional[torch.Tensor] = None
    camera_indices: Optional[torch.Tensor] = None
    nears: Optional[torch.Tensor] = None
    fars: Optional[torch.Tensor] = None
    metadata: Dict[str, torch.Tensor] = field(default_factory=dict)

    def __len__(self):
        return self.origins.shape[0]

    @property
    def shape(self):
        return self.origins.shape[:-1]

    def slice(self, start: int, end: int) -> "RayBundle":
        """row-major slice of a flat bundle (get_row_major_sliced_ray_bundle, neusky_model.py:1416)"""
        f = lambda t: None if t is None else t.reshape(-1, t.shape[-1])[start:end]
        return RayBundle(f(self.origins), f(self.directions), f(self.pixel_area), f(self.camera_indices), f(self.nears),
                         f(self.fars), {k: f(v) for k, v in self.metadata.items()})
