"""Host-side geometry of the multiresolution hash grid (tiny-cuda-nn HashGrid layout).

Mirrors the encoding_config dicts the reference hands to tcnn.Encoding
(neusky/fields/sdf_albedo_field.py:115-130, neusky/fields/directional_distance_field.py:139-156):
per level `scale = base * growth^l - 1`, `resolution = ceil(scale) + 1`, rows = min(round_up8(res^3), 2^log2T);
levels are concatenated in one [n_params, 2] float32 table (level-major, so a wave walking one level
touches one contiguous slab of HBM / Infinity Cache).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np


@dataclass
class HashGridGeometry:
    n_levels: int = 16
    log2_hashmap_size: int = 19
    base_res: int = 16
    max_res: int = 2048
    smoothstep: bool = False
    n_features: int = 2
    scales: List[float] = field(default_factory=list)
    resolutions: List[int] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)

    def __post_init__(self):
        if self.n_features != 2:
            raise NotImplementedError("the HIP hash-grid kernels are specialised for 2 features per level")
        if not 1 <= self.n_levels <= 16:
            raise ValueError("n_levels must be in [1, 16]")
        growth = np.exp((np.log(self.max_res) - np.log(self.base_res)) / (self.n_levels - 1)) if self.n_levels > 1 else 1.0
        log2_growth = np.float32(np.log2(np.float32(growth)))
        off = 0
        self.scales, self.resolutions, self.offsets = [], [], []
        for lvl in range(self.n_levels):
            scale = np.float32(np.exp2(np.float32(lvl) * log2_growth) * np.float32(self.base_res) - np.float32(1.0))
            res = int(np.ceil(scale)) + 1
            n = min((min(res**3, 2**31 - 1) + 7) // 8 * 8, 1 << self.log2_hashmap_size)
            self.scales.append(float(scale))
            self.resolutions.append(res)
            self.offsets.append(off)
            off += n
        self.offsets.append(off)

    @property
    def n_params(self) -> int:
        return self.offsets[-1]

    @property
    def out_dim(self) -> int:
        return self.n_levels * self.n_features
