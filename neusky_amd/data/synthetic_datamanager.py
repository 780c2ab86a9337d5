"""Synthetic NeRF-OSR-lk2-shaped ray batches (SURVEY.md section 8(d) 'Synthetic inputs').

Stands where `NeuSkyDataManager.next_train` / `get_sky_ray_bundle` stand in the reference
(neusky/data/datamanagers/neusky_datamanager.py:277-288, called neusky/pipelines/neusky_pipeline.py:252,503):
same output contract (RayBundle + batch{image [R,3], mask [R,4], indices}), data drawn from a seeded
generator instead of the NeRF-OSR image stack (there is no dataset in the build image).  Real
data parsers are SURVEY.md section 8(f) items 1 and 4.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Tuple, Type

import torch

from ..cameras.rays import RayBundle
from ..utils.utils import to_device_async
from ..plugin import ConfigBase


@dataclass
class SyntheticDataManagerConfig(ConfigBase):
    _target: Type = field(default_factory=lambda: SyntheticDataManager)
    num_train_images: int = 300
    num_eval_images: int = 96
    train_num_rays_per_batch: int = 1024
    image_height: int = 823
    image_width: int = 1280
    focal: float = 1100.0
    eval_image_height: int = 24   # synthetic eval frames are small: a full 823 x 1280 frame is BASELINE config 5's job
    eval_image_width: int = 32
    eval_num_rays_per_batch: int = 1024
    seed: int = 0
    camera_optimizer = None

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class _Dataset:
    def __init__(self, n, scene_box):
        self._n, self.scene_box, self.metadata = n, scene_box, {}

    def __len__(self):
        return self._n


class SyntheticDataManager:
    def __init__(self, config: SyntheticDataManagerConfig, device="cuda:0", test_mode="val", world_size=1, local_rank=0, **_):
        self.config, self.device, self.world_size, self.local_rank = config, device, world_size, local_rank
        scene_box = {"aabb": torch.tensor([[-1.0, -1.0, -1.0], [1.0, 1.0, 1.0]])}  # scene_scale = 1.0 (neusky_config.py:52)
        self.train_dataset = _Dataset(config.num_train_images, scene_box)
        self.eval_dataset = _Dataset(config.num_eval_images, scene_box)
        self.num_val = self.num_test = config.num_eval_images
        g = torch.Generator().manual_seed(config.seed)
        n = config.num_train_images
        # camera centres in the disc |xy| <= 0.8, z in [-0.05, 0.05] (poses auto-scaled to the unit box,
        # nerfosr_cityscapes_dataparser.py:272-279), looking at the origin with jitter
        ang = torch.rand(n, generator=g) * 2 * torch.pi
        rad = 0.8 * torch.sqrt(torch.rand(n, generator=g))
        self.cam_pos = torch.stack([rad * torch.cos(ang), rad * torch.sin(ang), (torch.rand(n, generator=g) - 0.5) * 0.1], 1)
        fwd = -self.cam_pos + 0.2 * torch.randn(n, 3, generator=g)
        fwd = fwd / fwd.norm(dim=-1, keepdim=True)
        up = torch.tensor([0.0, 0.0, 1.0]).expand(n, 3)
        right = torch.linalg.cross(fwd, up, dim=-1)
        right = right / right.norm(dim=-1, keepdim=True)
        self.cam_R = torch.stack([right, torch.linalg.cross(right, fwd, dim=-1), fwd], -1)  # columns: right, up', forward
        self._gen = torch.Generator().manual_seed(config.seed + 1 + local_rank)

    def get_param_groups(self) -> Dict:
        return {}

    def _rays(self, R: int, g: torch.Generator) -> Tuple[RayBundle, torch.Tensor]:
        c = self.config
        cam = torch.randint(0, c.num_train_images, (R,), generator=g)
        px = (torch.rand(R, generator=g) - 0.5) * c.image_width / c.focal
        py = (torch.rand(R, generator=g) - 0.5) * c.image_height / c.focal
        d_cam = torch.stack([px, py, torch.ones(R)], -1)
        d = torch.einsum("rij,rj->ri", self.cam_R[cam], d_cam)
        norm = d.norm(dim=-1, keepdim=True)
        dev = self.device
        rb = RayBundle(origins=to_device_async(self.cam_pos[cam], dev), directions=to_device_async(d / norm, dev),
                       pixel_area=torch.ones(R, 1, device=dev), camera_indices=to_device_async(cam[:, None], dev), metadata={"directions_norm": torch.ones(R, 1, device=dev)})
        return rb, cam

    def next_train(self, step: int):
        g = self._gen
        R = self.config.train_num_rays_per_batch
        rb, cam = self._rays(R, g)
        image = torch.rand(R, 3, generator=g)
        u = torch.rand(R, 4, generator=g)
        mask = torch.stack([u[:, 0] < 0.9, u[:, 1] < 0.6, u[:, 2] < 0.15, u[:, 3] < 0.3], -1)  # [static, fg, ground, sky]
        mask[:, 1] &= ~mask[:, 3]
        batch = {"image": to_device_async(image, self.device), "mask": to_device_async(mask, self.device), "indices": torch.stack([cam, cam * 0, cam * 0], 1)}
        return rb, batch

    def get_sky_ray_bundle(self, number_of_rays: int) -> RayBundle:
        g = self._gen
        rb, _ = self._rays(number_of_rays, g)
        d = rb.directions.clone()
        d[:, 2] = d[:, 2].abs() + 0.2  # sky rays point upwards (device-side ops, no host round trip)
        rb.directions = d / d.norm(dim=-1, keepdim=True)
        return rb

    # ------------------------------------------------------------------ evaluation side (NeuSkyDataManager.next_eval /
    # next_eval_image / get_eval_image_half_bundle, neusky/data/datamanagers/neusky_datamanager.py:236-333)
    class _EvalLoader:
        def __init__(self, n):
            self.image_indices = list(range(n))

        def __len__(self):
            return len(self.image_indices)

    @property
    def eval_dataloader(self):
        return SyntheticDataManager._EvalLoader(self.config.num_eval_images)

    def _eval_frame(self, image_idx: int):
        """synthetic eval frame `image_idx`: pinhole rays of train camera `image_idx % num_train`, seeded pixel colours / masks"""
        c = self.config
        H, W = c.eval_image_height, c.eval_image_width
        cam = image_idx % c.num_train_images
        g = torch.Generator().manual_seed(c.seed + 7919 * (image_idx + 1))
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        f = c.focal * W / c.image_width
        d_cam = torch.stack([(xs + 0.5 - W / 2) / f, (ys + 0.5 - H / 2) / f, torch.ones(H, W)], -1)
        d = torch.einsum("ij,hwj->hwi", self.cam_R[cam], d_cam)
        norm = d.norm(dim=-1, keepdim=True)
        image = torch.rand(H, W, 3, generator=g)
        u = torch.rand(H, W, 4, generator=g)
        mask = torch.stack([u[..., 0] < 0.9, u[..., 1] < 0.6, u[..., 2] < 0.15, u[..., 3] < 0.3], -1)
        mask[..., 1] &= ~mask[..., 3]
        return cam, d / norm, norm, image, mask

    def next_eval_image(self, idx: int):
        """-> (image_idx, camera_ray_bundle [H,W], batch{image [H,W,3], mask [H,W,4], image_idx})"""
        image_idx = int(idx) % self.config.num_eval_images
        cam, d, norm, image, mask = self._eval_frame(image_idx)
        H, W = d.shape[:2]
        dev = self.device
        rb = RayBundle(origins=to_device_async(self.cam_pos[cam].expand(H, W, 3).contiguous(), dev), directions=to_device_async(d.contiguous(), dev),
                       pixel_area=torch.ones(H, W, 1, device=dev), camera_indices=torch.full((H, W, 1), image_idx, dtype=torch.long, device=dev),
                       metadata={"directions_norm": to_device_async(norm.contiguous(), dev)})
        return image_idx, rb, {"image": to_device_async(image, dev), "mask": to_device_async(mask, dev), "image_idx": image_idx}

    def _eval_rays(self, image_idx: int, sample_region: str, n: int, g: torch.Generator):
        cam, d, norm, image, mask = self._eval_frame(image_idx)
        H, W = d.shape[:2]
        ok = mask[..., 0].clone()  # static mask, restricted to an image half (neusky_pixel_sampler.py:128-146)
        if sample_region == "left_image_half":
            ok[:, W // 2:] = False
        elif sample_region == "right_image_half":
            ok[:, :W // 2] = False
        pix = torch.nonzero(ok)
        pick = pix[torch.randint(0, pix.shape[0], (n,), generator=g)]
        y, x = pick[:, 0], pick[:, 1]
        dev = self.device
        rb = RayBundle(origins=to_device_async(self.cam_pos[cam].expand(n, 3).contiguous(), dev), directions=to_device_async(d[y, x].contiguous(), dev),
                       pixel_area=torch.ones(n, 1, device=dev), camera_indices=torch.full((n, 1), image_idx, dtype=torch.long, device=dev),
                       metadata={"directions_norm": to_device_async(norm[y, x].contiguous(), dev)})
        batch = {"image": to_device_async(image[y, x], dev), "mask": to_device_async(mask[y, x], dev),
                 "indices": torch.stack([torch.full_like(y, image_idx), y, x], 1)}
        return rb, batch

    def state_dict(self):
        """exact resume (utils.checkpoints): the host generator and the eval cursors"""
        return {"gen": self._gen.get_state(), "eval_cursor": getattr(self, "_eval_cursor", -1), "half_cursor": getattr(self, "_half_cursor", -1)}

    def load_state_dict(self, state):
        self._gen.set_state(state["gen"])
        self._eval_cursor, self._half_cursor = int(state.get("eval_cursor", -1)), int(state.get("half_cursor", -1))

    def next_eval(self, step: int):
        self._eval_cursor = (getattr(self, "_eval_cursor", -1) + 1) % self.config.num_eval_images
        return self._eval_rays(self._eval_cursor, "full_image", self.config.eval_num_rays_per_batch, self._gen)

    def get_eval_image_half_bundle(self, sample_region: str = "full_image", image_index=None, num_rays=None):
        if image_index is None:  # the reference cycles its eval image dataloader (:288-291)
            self._half_cursor = (getattr(self, "_half_cursor", -1) + 1) % self.config.num_eval_images
            image_index = self._half_cursor
        return self._eval_rays(int(image_index), sample_region, num_rays or self.config.eval_num_rays_per_batch, self._gen)
