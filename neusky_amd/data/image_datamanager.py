"""Device-resident image datamanager: ray generation + pixel / mask sampling on the GPU.

Takes the place of `NeuSkyDataManager.next_train / get_sky_ray_bundle` + `NeuSkyPixelSampler`
(neusky/data/datamanagers/neusky_datamanager.py:277-288, neusky/data/neusky_pixel_sampler.py:36-81) for an image stack
that is already on the device (`images_on_gpu=True, masks_on_gpu=True`, neusky_config.py:60-61):
  * masks have 4 channels [static, fg, ground, sky] (neusky/data/datasets/neusky_dataset.py:290); training pixels are
    drawn uniformly from the pixels whose channel 0 (static) is set (neusky_pixel_sampler.py:36-46);
  * sky rays are drawn from the pixels whose fg channel (1) is NOT set (`1 - mask[..., 1:2]`, :58-62);
  * rays come from pinhole cameras in nerfstudio's convention (x right, y up, camera looks along -z; pixel centres at
    +0.5), directions normalised, `metadata["directions_norm"]` = the norm before normalisation.
The reference round-trips the sampled indices through the CPU every step (:55-57, datamanager :286); here the valid-pixel
lists are built once and a step is three `randint` + gathers on the device (static shapes, hipGraph-safe).
Parsing NeRF-OSR / Cityscapes folders from disk is `neusky_amd.data.dataparsers`;
`DeviceImageDataManager.from_dataset` uploads what it reads.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from ..cameras.rays import RayBundle


class _Dataset:
    def __init__(self, n, scene_box):
        self._n, self.scene_box, self.metadata = n, scene_box, {}

    def __len__(self):
        return self._n


class DeviceImageDataManager:
    def __init__(self, images: torch.Tensor, masks: torch.Tensor, c2w: torch.Tensor, fx, fy, cx, cy,
                 train_num_rays_per_batch: int = 1024, device="cuda:0", scene_scale: float = 1.0, num_eval: int = 0, seed: int = 0,
                 image_idx: Optional[torch.Tensor] = None):
        """images [N,H,W,3] float, masks [N,H,W,4] bool, c2w [N,3,4] camera-to-world (nerfstudio/OpenGL convention);
        fx, fy, cx, cy: one pinhole for every image (floats) or per-image tensors [N]"""
        assert images.shape[:3] == masks.shape[:3] and masks.shape[-1] == 4 and c2w.shape[1:] == (3, 4)
        self.device = device
        self.images, self.masks, self.c2w = images.to(device), masks.to(device).bool(), c2w.to(device).float()
        self.N, self.H, self.W = images.shape[:3]
        per_image = lambda v: torch.as_tensor(v, dtype=torch.float32).reshape(-1).expand(self.N).contiguous().to(device)  # noqa: E731
        self.fx, self.fy, self.cx, self.cy = per_image(fx), per_image(fy), per_image(cx), per_image(cy)
        self.train_num_rays_per_batch = train_num_rays_per_batch
        # dataset index of every stacked image: `indices[:, 0] = batch["image_idx"][c]` (neusky_pixel_sampler.py:76,155)
        self.image_idx = (torch.arange(self.N) if image_idx is None else torch.as_tensor(image_idx)).to(device).long()
        scene_box = {"aabb": torch.tensor([[-scene_scale] * 3, [scene_scale] * 3])}
        self.train_dataset = _Dataset(self.N, scene_box)
        self.eval_dataset = _Dataset(max(num_eval, 1), scene_box)
        self.num_val = self.num_test = max(num_eval, 1)
        # valid-pixel lists, built once (the only nonzero() calls; never inside a step)
        self.static_pixels = torch.nonzero(self.masks[..., 0])        # [K,3] = (image, y, x)
        self.sky_pixels = torch.nonzero(~self.masks[..., 1])          # 1 - fg
        assert self.static_pixels.shape[0] > 0, "no pixel with the static mask set"
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self._half_pools: Dict = {}

    @classmethod
    def from_dataset(cls, dataset, **kw) -> "DeviceImageDataManager":
        """upload a parsed on-disk dataset (`neusky_amd.data.dataparsers.NeuSkyDataset`; frames cropped or padded to
        one size) with its per-image intrinsics, as `images_on_gpu / masks_on_gpu` does (neusky_config.py:60-61)"""
        from .dataparsers import load_stacks
        images, masks = load_stacks(dataset)
        cams = dataset.cameras
        s = float(dataset.scale_factor)
        kw.setdefault("scene_scale", float(dataset.scene_box["aabb"][1, 0]))
        return cls(images, masks, cams.camera_to_worlds, cams.fx * s, cams.fy * s, cams.cx * s, cams.cy * s, **kw)

    def get_param_groups(self) -> Dict:
        return {}

    def generate_rays(self, indices: torch.Tensor) -> RayBundle:
        """indices [R,3] = (image, y, x) on the device -> RayBundle (pinhole, nerfstudio convention)"""
        c, y, x = indices[:, 0], indices[:, 1].float(), indices[:, 2].float()
        fx, fy = self.fx[c], self.fy[c]
        d_cam = torch.stack([(x + 0.5 - self.cx[c]) / fx, -(y + 0.5 - self.cy[c]) / fy, -torch.ones_like(x)], -1)
        R = self.c2w[c, :, :3]
        d = torch.einsum("rij,rj->ri", R, d_cam)
        norm = d.norm(dim=-1, keepdim=True)
        n = indices.shape[0]
        return RayBundle(origins=self.c2w[c, :, 3].contiguous(), directions=(d / norm).contiguous(),
                         pixel_area=(1.0 / (fx * fy))[:, None],
                         camera_indices=c[:, None].contiguous(), metadata={"directions_norm": norm})

    def _draw(self, pixels: torch.Tensor, n: int) -> torch.Tensor:
        pick = torch.randint(0, pixels.shape[0], (n,), device=pixels.device, generator=self.gen)
        return pixels[pick]

    def collate(self, idx: torch.Tensor) -> Dict:
        """batch of the stacked pixels idx [R,3] = (stack position, y, x): values gathered by [c, y, x], first index column
        remapped to the dataset's image index (neusky_pixel_sampler.py:71-77)"""
        c, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
        out = idx.clone()
        out[:, 0] = self.image_idx[c]
        return {"image": self.images[c, y, x], "mask": self.masks[c, y, x], "indices": out}

    def half_pixels(self, sample_region: str = "full_image", image_index: Optional[int] = None) -> torch.Tensor:
        """pixels admissible for eval-latent fitting: static mask (channel 0) restricted to an image half
        (neusky_pixel_sampler.py:128-146); all stacked images, or one"""
        sel = self.static_pixels if image_index is None else self.static_pixels[self.static_pixels[:, 0] == image_index]
        if sample_region == "left_image_half":
            sel = sel[sel[:, 2] < self.W // 2]
        elif sample_region == "right_image_half":
            sel = sel[sel[:, 2] >= self.W // 2]
        return sel

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        idx = self._draw(self.static_pixels, self.train_num_rays_per_batch)
        return self.generate_rays(idx), self.collate(idx)

    def get_sky_ray_bundle(self, number_of_rays: int) -> RayBundle:
        pool = self.sky_pixels if self.sky_pixels.shape[0] > 0 else self.static_pixels
        return self.generate_rays(self._draw(pool, number_of_rays))

    def get_eval_image_half_bundle(self, sample_region: str = "full_image", image_index: Optional[int] = None, num_rays: Optional[int] = None):
        """rays restricted to an image half and to the static mask, for eval-latent fitting: drawn over ALL stacked images (the
        reference's eval pixel sampler works on its cached eval image batch, datamanager :288-305) or over one (`image_index`)"""
        n = num_rays or self.train_num_rays_per_batch
        key = (sample_region, image_index)
        pool = self._half_pools.get(key)
        if pool is None:  # built once per (region, image): nonzero / boolean indexing never runs inside a replayed step
            pool = self._half_pools[key] = self.half_pixels(sample_region, image_index)
        idx = self._draw(pool, n)
        return self.generate_rays(idx), self.collate(idx)

    def pixels_of_images(self, positions) -> torch.Tensor:
        """static-mask pixels of the stacked images `positions` (a list of stack positions)"""
        sel = torch.zeros(self.N, dtype=torch.bool, device=self.device)
        sel[torch.as_tensor(list(positions), dtype=torch.long, device=self.device)] = True
        return self.static_pixels[sel[self.static_pixels[:, 0]]]

    def state_dict(self) -> Dict:
        """what an exact resume needs: the generator's state (utils.checkpoints)"""
        return {"gen": self.gen.get_state()}

    def load_state_dict(self, state: Dict) -> None:
        self.gen.set_state(state["gen"].cpu())


# =====================================================================================================================
# The reference's datamanager seam (neusky/data/datamanagers/neusky_datamanager.py:56-288): a config with a dataparser, train / eval
# datasets parsed from disk, every image and mask resident on the device (`images_on_gpu / masks_on_gpu`, neusky_config.py:60-61), and
# the iterator functions the pipeline calls.  Built on DeviceImageDataManager: one instance for the train split, one for the eval split.
from dataclasses import dataclass, field  # noqa: E402
from pathlib import Path  # noqa: E402
from typing import Any, Type  # noqa: E402

from ..plugin import ConfigBase  # noqa: E402


@dataclass
class NeuSkyDataManagerConfig(ConfigBase):
    """field names of the reference's NeuSkyDataManagerConfig / nerfstudio VanillaDataManagerConfig that the `neusky` method sets
    (neusky_config.py:46-64)"""
    _target: Type = field(default_factory=lambda: NeuSkyDataManager)
    dataparser: Any = None                      # NeRFOSRCityScapesDataParserConfig / CustomNeuskyDataparserConfig (data/dataparsers.py)
    data: Optional[Path] = None                 # overrides dataparser.data when set (ns-train --data)
    train_num_rays_per_batch: int = 1024
    eval_num_rays_per_batch: int = 1024
    train_num_images_to_sample_from: int = -1   # -1: all images resident (the only mode built)
    train_num_times_to_repeat_images: int = -1
    images_on_gpu: bool = True
    masks_on_gpu: bool = True
    camera_res_scale_factor: float = 1.0

    def setup(self, **kwargs):
        return self._target(self, **kwargs)


class NeuSkyDataManager:
    """train / eval data of one scene on the device.  Surface used by NeuSkyPipeline (neusky_pipeline.py:146-148,241-291,356-455):
    train_dataset / eval_dataset (len, scene_box, metadata), num_val / num_test, next_train, next_eval, next_eval_image,
    eval_dataloader, get_sky_ray_bundle, get_eval_image_half_bundle, get_param_groups."""

    def __init__(self, config: NeuSkyDataManagerConfig, device="cuda:0", test_mode: str = "val", world_size: int = 1, local_rank: int = 0,
                 eval_latent_optimise_method: Optional[str] = "per_image", **_):
        from .dataparsers import NeRFOSRCityScapesDataParserConfig, NeuSkyDataset
        if config.train_num_images_to_sample_from != -1:
            raise NotImplementedError("train_num_images_to_sample_from: every image is resident on the device (images_on_gpu)")
        self.config, self.device, self.test_mode = config, device, test_mode
        self.eval_latent_optimise_method = eval_latent_optimise_method or "per_image"  # passed by the pipeline (neusky_pipeline.py:129-135)
        pc = config.dataparser if config.dataparser is not None else NeRFOSRCityScapesDataParserConfig()
        if config.data is not None:
            pc.data = Path(config.data)
        self.dataparser = pc.setup()
        eval_split = "test" if test_mode in ("test", "inference") else "val"  # datamanager :97
        s = float(config.camera_res_scale_factor)
        self.train_dataset = NeuSkyDataset(self.dataparser.get_dataparser_outputs(split="train"), scale_factor=s, split="train")
        self.eval_dataset = NeuSkyDataset(self.dataparser.get_dataparser_outputs(split=eval_split), scale_factor=s, split=eval_split)
        # every rank holds the whole scene and draws its own rays (ray-sharded data parallelism): the generators are seeded by the GLOBAL
        # rank (two nodes' ranks with the same local rank must not draw the same batches)
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else local_rank
        self.train = DeviceImageDataManager.from_dataset(self.train_dataset, train_num_rays_per_batch=config.train_num_rays_per_batch,
                                                         device=device, seed=rank)
        self.eval = DeviceImageDataManager.from_dataset(self.eval_dataset, train_num_rays_per_batch=config.eval_num_rays_per_batch,
                                                        device=device, seed=100_003 + rank)
        md = self.eval_dataset.metadata
        if self.eval_latent_optimise_method == "per_image":  # datamanager :114-119: one latent per image
            self.num_test = len(self.eval_dataset) if eval_split == "test" else len(self.dataparser.get_dataparser_outputs(split="test").image_filenames)
            self.num_val = len(self.eval_dataset)
            self.indices_to_session = None
        else:
            # NeRF-OSR relighting benchmark (:120-122,183-233): one latent per capture SESSION; the latents are fitted on one held-out image
            # per session and compared on the images that have an evaluation mask
            if md.get("session_to_indices") is None:
                raise ValueError(f"eval_latent_optimise_method={self.eval_latent_optimise_method!r} needs a dataparser with session metadata")
            s2i = md["session_to_indices"]
            self.indices_to_session = md["indices_to_session"]
            self.num_test = self.num_val = len(s2i)
            self.holdout_indices = [s2i[k][rel] for k, rel in zip(s2i.keys(), md["session_holdout_indices"])]
            self.compare_indices = list(self.eval_dataset.test_eval_mask_dict.keys())
            self._session_of = torch.tensor([self.indices_to_session[i] for i in range(len(self.eval_dataset))], dtype=torch.long, device=device)
            self._session_pools = {"optimise": self.eval.pixels_of_images(self.holdout_indices),
                                   "compare": self.eval.pixels_of_images(self.compare_indices or self.holdout_indices)}
            self._compare_cursor = 0

    def get_param_groups(self) -> Dict:
        return {}

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        return self.train.next_train(step)

    def get_sky_ray_bundle(self, number_of_rays: int) -> RayBundle:
        return self.train.get_sky_ray_bundle(number_of_rays)

    def next_eval(self, step: int) -> Tuple[RayBundle, Dict]:
        return self.eval.next_train(step)

    def get_eval_image_half_bundle(self, sample_region: str = "full_image", image_index: Optional[int] = None, num_rays: Optional[int] = None):
        """datamanager :288-305: rays of the eval images (all of them unless `image_index` names one) inside an image half and the
        static mask; batch["indices"][:, 0] = the eval image = the row of the eval latent table the ray trains"""
        return self.eval.get_eval_image_half_bundle(sample_region, image_index, num_rays)

    def get_nerfosr_lighting_eval_bundle(self, stage: str):
        """datamanager :307-330: rays of the held-out image of every session ("optimise") or of the images with an evaluation mask
        ("compare"), static mask; the image index of every ray is replaced by its SESSION index (one RENI++ latent per session).
        The reference generates these rays with its TRAIN ray generator (:321); the eval cameras are used here."""
        assert stage in ("optimise", "compare")
        if self.indices_to_session is None:
            raise ValueError("get_nerfosr_lighting_eval_bundle: the datamanager was set up with eval_latent_optimise_method='per_image'")
        e = self.eval
        idx = e._draw(self._session_pools[stage], e.train_num_rays_per_batch)
        rb, batch = e.generate_rays(idx), e.collate(idx)
        sess = self._session_of[batch["indices"][:, 0]]
        batch["indices"][:, 0] = sess
        rb.camera_indices = sess[:, None].contiguous()
        return rb, batch

    def state_dict(self) -> Dict:
        return {"train": self.train.state_dict(), "eval": self.eval.state_dict()}

    def load_state_dict(self, state: Dict) -> None:
        self.train.load_state_dict(state["train"])
        self.eval.load_state_dict(state["eval"])

    def next_eval_image(self, idx: int):
        """-> (image_idx, camera ray bundle [H, W], batch {image [H,W,3], mask [H,W,4], image_idx}); in the NeRF-OSR session modes
        the images are the ones with an evaluation mask and image_idx is the SESSION of the image (:239-253)"""
        e = self.eval
        if self.indices_to_session is not None and self.compare_indices:
            i = self.compare_indices[int(idx) % len(self.compare_indices)]
        else:
            i = int(idx) % e.N
        yy, xx = torch.meshgrid(torch.arange(e.H, device=e.device), torch.arange(e.W, device=e.device), indexing="ij")
        pix = torch.stack([torch.full_like(yy, i), yy, xx], -1).reshape(-1, 3)
        rb = e.generate_rays(pix)
        shape = lambda t: t.reshape(e.H, e.W, *t.shape[1:])  # noqa: E731
        bundle = RayBundle(origins=shape(rb.origins), directions=shape(rb.directions), pixel_area=shape(rb.pixel_area),
                           camera_indices=shape(rb.camera_indices), metadata={"directions_norm": shape(rb.metadata["directions_norm"])})
        image_idx = int(e.image_idx[i])
        if self.indices_to_session is not None:
            image_idx = int(self.indices_to_session[image_idx])
            bundle.camera_indices = torch.full_like(bundle.camera_indices, image_idx)
        return image_idx, bundle, {"image": e.images[i], "mask": e.masks[i], "image_idx": image_idx}

    class _EvalLoader:
        def __init__(self, dm):
            self.dm = dm

        def __len__(self):
            return self.dm.eval.N

        def __iter__(self):
            for i in range(self.dm.eval.N):
                _, rb, batch = self.dm.next_eval_image(i)
                yield rb, batch

    @property
    def eval_dataloader(self):
        return NeuSkyDataManager._EvalLoader(self)
