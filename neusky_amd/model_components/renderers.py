"""Lambertian renderer with visibility on the fused HIP hemisphere kernel.

Mirrors `neusky.model_components.renderers.RGBLambertianRendererWithVisibility`
(neusky/model_components/renderers.py:56-176).  The reference signature takes light directions / colours
/ visibility broadcast to [R*S, D, *]; those broadcasts are what the kernel avoids, so the hot-path entry
point is `forward_compact`.  `forward` accepts the reference's broadcast layout, checks that it really is
a broadcast of compact data, and routes to the same kernel.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from .. import ops


class RGBLambertianRendererWithVisibility(nn.Module):
    def forward_compact(self, albedos, normals, light_directions, cam_colours, cam_of_ray, visibility,
                        background_illumination, weights) -> torch.Tensor:
        """albedos/normals [R,S,3]; light_directions [D,3]; cam_colours [U,D,3]; cam_of_ray [R] int32;
        visibility [R,D] or None; background_illumination [R,3]; weights [R,S,1] -> rgb [R,3]"""
        rgb = ops.HemiCompositeFn.apply(albedos, normals, weights[..., 0], light_directions, cam_colours, cam_of_ray,
                                        visibility, background_illumination)
        return rgb  # linear_to_sRGB already clamps to [0,1]; the eval-mode clamp (:173-174) is a no-op

    def forward(self, albedos, normals, light_directions, light_colors, visibility, background_illumination, weights,
                ray_indices: Optional[torch.Tensor] = None, num_rays: Optional[int] = None) -> torch.Tensor:
        if ray_indices is not None:
            raise NotImplementedError("packed samples (nerfacc branch, renderers.py:117-120) are never used by NeuSky")
        R, S = weights.shape[:2]
        D = light_directions.shape[-2]
        ld = light_directions.reshape(R, S, D, 3)
        lc = light_colors.reshape(R, S, D, 3)
        dirs = ld[0, 0]
        cols = lc[:, 0]
        if not (torch.equal(ld, dirs.expand_as(ld)) and torch.equal(lc, cols[:, None].expand_as(lc))):
            raise ValueError("light directions / colours are not a broadcast of [D,3] / per-ray [R,D,3] data")
        vis = None
        if visibility is not None:
            v = visibility.reshape(R, S, D)
            vis = v[:, 0]
            if not torch.equal(v, vis[:, None].expand_as(v)):
                raise ValueError("visibility is not constant along the samples of a ray")
        cam = torch.arange(R, dtype=torch.int32, device=weights.device)
        return self.forward_compact(albedos, normals, dirs.contiguous(), cols.contiguous(), cam, vis,
                                    background_illumination, weights)
