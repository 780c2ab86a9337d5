"""von Mises-Fisher DDF ray sampler (host side, like the reference: CPU RNG then `.to(device)`).
Mirrors neusky/model_components/ddf_sampler.py:183-286 (VMFDDFSampler)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Type

import torch

from ..cameras.rays import RayBundle
from ..utils.utils import device_rng, to_device_async
from ..plugin import ConfigBase


@dataclass
class VMFDDFSamplerConfig(ConfigBase):
    _target: Type = field(default_factory=lambda: VMFDDFSampler)
    num_samples_on_sphere: int = 8
    num_rays_per_sample: int = 128
    only_sample_upper_hemisphere: bool = True
    concentration: float = 20.0

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class VMFDDFSampler:
    def __init__(self, config: VMFDDFSamplerConfig, ddf_sphere_radius: float = 1.0, device="cpu"):
        self.config = config
        self.ddf_sphere_radius = ddf_sphere_radius
        self.device = device
        self.concentration = config.concentration

    def _random_vmf_cos(self, d: int, kappa: float, n: int, generator=None) -> torch.Tensor:
        """ddf_sampler.py:205-223 (Wood 1994 rejection sampler; Beta((d-1)/2,(d-1)/2) == U(0,1) for d = 3)"""
        b = (d - 1) / (2 * kappa + (4 * kappa**2 + (d - 1) ** 2) ** 0.5)
        x0 = (1 - b) / (1 + b)
        c = kappa * x0 + (d - 1) * math.log(1 - x0**2)
        out, found = [], 0
        beta = torch.distributions.beta.Beta((d - 1) / 2, (d - 1) / 2)
        while found < n:
            m = min(n, int((n - found) * 1.5))
            z = torch.rand(m, generator=generator) if d == 3 else beta.sample((m,))
            t = (1 - (1 + b) * z) / (1 - (1 - b) * z)
            test = kappa * t + (d - 1) * torch.log(1 - x0 * t) - c
            accept = test >= -math.e  # sic (:220)
            out.append(t[accept])
            found += int(accept.sum())
        return torch.cat(out)[:n]

    def random_vmf(self, normals: torch.Tensor, kappa: float, num_samples: int, generator=None) -> torch.Tensor:
        """ddf_sampler.py:225-247"""
        normals = normals / torch.norm(normals, dim=-1, keepdim=True)
        N, d = normals.shape
        z = torch.randn(N, num_samples, d, generator=generator)
        z = z / torch.norm(z, dim=-1, keepdim=True)
        z = z - torch.einsum("nij,nj->ni", z, normals)[..., None] * normals[:, None, :]
        z = z / torch.norm(z, dim=-1, keepdim=True)
        cos = self._random_vmf_cos(d, kappa, N * num_samples, generator).reshape(N, num_samples)
        sin = torch.sqrt(1 - cos**2)
        x = z * sin[..., None] + cos[..., None] * normals[:, None, :]
        return x / torch.norm(x, dim=-1, keepdim=True)

    def generate_ddf_samples(self, num_positions: int, num_directions: int, positions: Optional[torch.Tensor] = None,
                             generator=None) -> RayBundle:
        """ddf_sampler.py:249-286"""
        if positions is None:
            theta = 2 * torch.pi * torch.rand(num_positions, generator=generator)
            phi = torch.acos(2 * torch.rand(num_positions, generator=generator) - 1)
            positions = torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], 1)
        if self.config.only_sample_upper_hemisphere:
            positions = torch.where(positions[:, 2:3] < 0, -positions, positions)
        directions = self.random_vmf(-positions, self.concentration, num_directions, generator)
        flip = torch.einsum("nij,nj->ni", directions, -positions) < 0
        directions = torch.where(flip[..., None], -directions, directions)
        positions = positions * self.ddf_sphere_radius
        positions = to_device_async(positions.unsqueeze(1).repeat(1, num_directions, 1).reshape(-1, 3), self.device)
        directions = to_device_async(directions.reshape(-1, 3), self.device)
        n = positions.shape[0]
        return RayBundle(origins=positions, directions=directions, pixel_area=torch.ones(n, 1, device=self.device),
                         camera_indices=torch.zeros(n, 1, device=self.device, dtype=torch.int64),
                         metadata={"directions_norm": torch.ones(n, 1, device=self.device)})

    def generate_ddf_samples_device(self, num_positions: int, num_directions: int) -> RayBundle:
        """Same distribution as generate_ddf_samples, drawn by ONE HIP kernel (csrc/samplers.hip: counter-based RNG, the
        rejection loop of :215-223 per sample, no host RNG, no upload, hipGraph-capturable: the call counter lives on the device)."""
        from .. import hip
        dev = torch.device(self.device)
        n = num_positions * num_directions
        st = getattr(self, "_dev_state", None)
        if st is None or st["n"] != n or st["ones"].device != dev:
            st = self._dev_state = {"n": n, "ones": torch.ones(n, 1, device=dev), "norm": torch.ones(n, 1, device=dev),
                                    "cam": torch.zeros(n, 1, device=dev, dtype=torch.int64)}
        origins = torch.empty(n, 3, device=dev)
        directions = torch.empty(n, 3, device=dev)
        seed, counter = device_rng(self, "ddf_vmf_samples", 0, dev)  # (registered: a checkpoint stores and restores the call counter)
        hip.ddf_vmf_samples(num_positions, num_directions, self.concentration, self.ddf_sphere_radius,
                            self.config.only_sample_upper_hemisphere, seed, counter, origins, directions)
        return RayBundle(origins=origins, directions=directions, pixel_area=st["ones"], camera_indices=st["cam"],
                         metadata={"directions_norm": st["norm"]})

    def __call__(self, generator=None) -> RayBundle:
        if generator is None:  # (a caller-supplied host generator = the reference's own host draw, ddf_sampler.py:249-286)
            return self.generate_ddf_samples_device(self.config.num_samples_on_sphere, self.config.num_rays_per_sample)
        return self.generate_ddf_samples(self.config.num_samples_on_sphere, self.config.num_rays_per_sample, generator=generator)
