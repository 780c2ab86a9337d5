"""Golden vectors for the pixel / mask sampling semantics of the reference:
neusky/data/neusky_pixel_sampler.py:36-46 (training pixels come from mask channel 0), :58-81 (sky pixels come from
1 - mask channel 1; collation by [c, y, x]; indices[:, 0] remapped through image_idx), :128-160 (image halves under the
static mask), as driven by neusky/data/datamanagers/neusky_datamanager.py:277-307.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_sampler.py      # writes tests/golden/pixel_sampler.npz

nerfstudio's PixelSampler (the random draw itself) is absent; its `sample_method` is replaced by a deterministic stand-in
that ENUMERATES the pixels the mask admits, in order, so the fixture pins exactly what is in-tree: which pixels are
eligible in each mode and how a batch is collated."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stub_importer  # noqa: E402

_ref_stub_importer.install()
import nerfstudio.data.pixel_samplers as ps  # noqa: E402  (placeholder module)


def enumerate_mask(self, batch_size, num_images, image_height, image_width, mask=None, device="cpu"):
    """stand-in for nerfstudio PixelSampler.sample_method: the first batch_size admissible pixels, cyclically"""
    assert mask is not None and mask.shape[-1] == 1, "the in-tree override must hand over a 1-channel mask"
    pix = torch.nonzero(mask[..., 0] > 0)
    return pix[torch.arange(batch_size) % pix.shape[0]].clone()


ps.PixelSampler.sample_method = enumerate_mask
import neusky.data.neusky_pixel_sampler as rs  # noqa: E402


def main():
    g = np.random.default_rng(11)
    N, H, W, n = 3, 6, 8, 40
    image = g.uniform(0, 1, (N, H, W, 3)).astype(np.float32)
    u = g.uniform(0, 1, (N, H, W, 4))
    mask = np.stack([u[..., 0] < 0.8, u[..., 1] < 0.55, u[..., 2] < 0.2, u[..., 3] < 0.3], -1)
    mask[..., 1] &= ~mask[..., 3]
    mask_f = mask.astype(np.float32)
    sampler = object.__new__(rs.NeuSkyPixelSampler)
    batch = {"image": torch.from_numpy(image), "mask": torch.from_numpy(mask_f), "image_idx": torch.tensor([5, 2, 7])}
    out = {"image": image, "mask": mask, "image_idx": batch["image_idx"].numpy(), "n": np.array(n)}
    # training pixels: the override slices the 4-channel mask to channel 0 (:36-46)
    count = int(mask[..., 0].sum())
    out["train_pixels"] = sampler.sample_method(count, N, H, W, mask=batch["mask"], device="cpu").numpy()
    # sky rays (:58-81)
    sky_count = int((~mask[..., 1]).sum())
    full = sampler.collate_sky_ray_batch(dict(batch), num_rays_per_batch=sky_count)
    out["sky_pixels_remapped"] = full["indices"].numpy()
    part = sampler.collate_sky_ray_batch(dict(batch), num_rays_per_batch=n)
    out["sky_batch_indices"], out["sky_batch_image"], out["sky_batch_mask"] = part["indices"].numpy(), part["image"].numpy(), part["mask"].numpy()
    # image halves under the static mask (:128-160)
    for region in ("left_image_half", "right_image_half", "full_image"):
        m = mask[..., 0].copy()
        if region == "left_image_half":
            m[:, :, W // 2:] = False
        elif region == "right_image_half":
            m[:, :, :W // 2] = False
        full = sampler.collate_image_half(dict(batch), num_rays_per_batch=int(m.sum()), sample_region=region)
        out[f"{region}_pixels_remapped"] = full["indices"].numpy()
        part = sampler.collate_image_half(dict(batch), num_rays_per_batch=n, sample_region=region)
        out[f"{region}_batch_image"], out[f"{region}_batch_mask"] = part["image"].numpy(), part["mask"].numpy()
        out[f"{region}_batch_indices"] = part["indices"].numpy()
    path = os.path.join(HERE, "pixel_sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
