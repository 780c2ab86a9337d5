"""Golden vectors for the on-disk parsers, produced by RUNNING the reference's
`NeRFOSRCityScapes._generate_dataparser_outputs`, `CustomNeuskyDataparser._generate_dataparser_outputs` and the
`NeuSkyDataset` image / mask readers (imported read-only from /root/reference under the stub importer) on the seeded
fixtures of tests/golden/dataset_fixture.py:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_dataparser.py

nerfstudio is absent, so `camera_utils.auto_orient_and_center_poses` is replaced by the product's restatement
(neusky_amd/data/dataparsers.py) -- the vectors therefore pin everything the reference's OWN files do (file discovery and
ordering, pose convention flip, crop/pad principal points, z-shift, auto scale, split slicing, session maps, eval-mask and
object-mask matching, mask channel semantics, crop / pad / rescale of images and masks) and NOT that nerfstudio function.
Only the resulting .npz travels; the reference never does."""
from __future__ import annotations

import functools
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _ref_stub_importer  # noqa: E402

_ref_stub_importer.install()
import dataset_fixture as fx  # noqa: E402
import neusky.data.dataparsers.nerfosr_cityscapes_dataparser as rdp  # noqa: E402
import neusky.data.dataparsers.custom_neusky_dataparser as rcp  # noqa: E402
import neusky.data.datasets.neusky_dataset as rds  # noqa: E402
from neusky_amd.data import dataparsers as ours  # noqa: E402

NS = types.SimpleNamespace
_cam_utils = NS(auto_orient_and_center_poses=ours.auto_orient_and_center_poses)
rdp.camera_utils = _cam_utils
rcp.camera_utils = _cam_utils


class _Log:
    def log(self, *a, **k):
        pass

    print = log


rcp.CONSOLE = _Log()


def names(paths):
    return np.array([os.path.basename(p) if p is not None else "" for p in paths])


def osr_config(data, **over):
    base = dict(data=data, scene="site1", scene_scale=1.0, scale_factor=1.0, orientation_method="vertical", center_method="focus",
                auto_scale_poses=True, mask_source="cityscapes", crop_to_equal_size=True, pad_to_equal_size=False,
                run_segmentation_inference=False, mask_vegetation=True, session_holdout_indices=[0, 0, 0],
                mask_out_of_view_frustum_objects=True, include_sidewalk_in_ground_mask=True)
    base.update(over)
    return NS(**base)


def run_osr(cfg, split):
    self = NS(config=cfg, width_height=[])
    out = rdp.NeRFOSRCityScapes._generate_dataparser_outputs(self, split)
    return out


def dataset_self(out, split, scale_factor=1.0):
    md = out.metadata
    self = NS(_dataparser_outputs=out, metadata=md, semantics=md["semantics"], split=split, scale_factor=scale_factor,
              crop_to_equal_size=md["crop_to_equal_size"], pad_to_equal_size=md["pad_to_equal_size"],
              test_eval_mask_dict=md["test_eval_mask_dict"], out_of_view_frustum_objects_masks=md["out_of_view_frustum_objects_masks"])
    if md["crop_to_equal_size"]:
        self.min_width, self.min_height = md["width_height"]
    if md["pad_to_equal_size"]:
        self.max_width, self.max_height = md["width_height"]
    self.get_mask_from_semantics = functools.partial(rds.NeuSkyDataset.get_mask_from_semantics, self)
    return self


def pack_outputs(prefix, out, store):
    cam = out.cameras
    store[f"{prefix}_c2w"] = cam.camera_to_worlds.numpy()
    for k in ("fx", "fy", "cx", "cy"):
        store[f"{prefix}_{k}"] = getattr(cam, k).numpy()
    store[f"{prefix}_images"] = names(out.image_filenames)
    md = out.metadata
    store[f"{prefix}_object_masks"] = names(md["out_of_view_frustum_objects_masks"])
    store[f"{prefix}_width_height"] = np.array(md["width_height"], dtype=np.int64)
    if md["session_to_indices"] is not None:
        s2i = md["session_to_indices"]
        store[f"{prefix}_session_sizes"] = np.array([len(s2i[k]) for k in sorted(s2i)], dtype=np.int64)
        store[f"{prefix}_session_members"] = np.array([i for k in sorted(s2i) for i in s2i[k]], dtype=np.int64)
        i2s = md["indices_to_session"]
        store[f"{prefix}_index_session"] = np.array([i2s[i] for i in range(len(out.image_filenames))], dtype=np.int64)
    tk = sorted(md["test_eval_mask_dict"])
    store[f"{prefix}_eval_mask_idx"] = np.array(tk, dtype=np.int64)
    store[f"{prefix}_eval_mask_names"] = names([md["test_eval_mask_dict"][k] for k in tk])
    if md.get("semantics") is not None:
        store[f"{prefix}_semantic_files"] = names(md["semantics"].filenames)


def main():
    store = {}
    with tempfile.TemporaryDirectory() as tmp:
        data = fx.build_nerfosr(os.path.join(tmp, "osr"))
        # ---- NeRF-OSR, the configuration of neusky_config.py:47-56 (crop) for every split
        cfg = osr_config(data)
        for split in ("train", "val", "test"):
            out = run_osr(cfg, split)
            pack_outputs(f"osr_crop_{split}", out, store)
            ds = dataset_self(out, "validation" if split == "val" else split)
            for i in range(len(out.image_filenames)):
                store[f"osr_crop_{split}_img{i}"] = rds.NeuSkyDataset.get_numpy_image(ds, i)
                store[f"osr_crop_{split}_mask{i}"] = rds.NeuSkyDataset.get_mask(ds, i).numpy().astype(np.uint8)
            if split == "train":
                env = out.metadata["envmap_cameras"]
                store["osr_env_fx"], store["osr_env_cx"], store["osr_env_cy"] = env.fx.numpy(), env.cx.numpy(), env.cy.numpy()
                store["osr_env_c2w"] = env.camera_to_worlds.numpy()
                store["osr_env_files"] = np.array([os.path.relpath(p, data) for p in out.metadata["envmap_filenames"]])
                # (the reference's get_envmap reads `_dataparser_outputs.envmap_filenames`, an attribute the outputs never
                #  carry -- the list lives in metadata, neusky_dataset.py:342 vs dataparser :450 -- so it is not callable)
        # ---- pad to equal size, vegetation kept as foreground, no sidewalk in the ground mask, half resolution
        cfg = osr_config(data, crop_to_equal_size=False, pad_to_equal_size=True, mask_vegetation=False,
                         include_sidewalk_in_ground_mask=False, mask_out_of_view_frustum_objects=False,
                         orientation_method="up", center_method="poses", scale_factor=0.5)
        out = run_osr(cfg, "train")
        pack_outputs("osr_pad_train", out, store)
        ds = dataset_self(out, "train", scale_factor=0.5)
        for i in (0, 3):
            store[f"osr_pad_train_img{i}"] = rds.NeuSkyDataset.get_numpy_image(ds, i)
            store[f"osr_pad_train_mask{i}"] = rds.NeuSkyDataset.get_mask(ds, i).numpy().astype(np.uint8)
        # ---- synthetic layout
        scene = fx.build_synthetic(os.path.join(tmp, "syn"))
        for tag, over in (("syn", {}), ("synsfm", {"center_method_sfm": True})):
            c = dict(data=Path(scene), transforms_filename="transforms.json", scene_scale=1.0, scale_factor=1.0,
                     orientation_method="vertical", center_method="focus", auto_scale_poses=True, mask_vegetation=False,
                     include_sidewalk_in_ground_mask=True, center_method_sfm=False, sfm_outlier_percentile=95.0,
                     sfm_scale_percentile=50.0, sfm_target_radius=0.5, points3d_filename="points3d.ply")
            c.update(over)
            self = NS(config=NS(**c))
            for m in ("_load_transforms", "_get_split_files", "_discover_gt_layers", "_resolve_gt_envmaps", "_load_sfm_points",
                      "_load_ply_numpy", "_compute_sfm_centering"):
                setattr(self, m, functools.partial(getattr(rcp.CustomNeuskyDataparser, m), self))
            for split in ("train", "val", "test"):
                out = rcp.CustomNeuskyDataparser._generate_dataparser_outputs(self, split)
                pack_outputs(f"{tag}_{split}", out, store)
                store[f"{tag}_{split}_orientation"] = out.metadata["orientation_rotation"].numpy()
                store[f"{tag}_{split}_gt_keys"] = np.array(sorted(k for k in out.metadata if k.startswith("gt_") and k.endswith("_filenames")))
                store[f"{tag}_{split}_envmap_hits"] = np.array([(-1.0 if e is None else float(e["rotation"])) for e in out.metadata["gt_envmap_info"]])
    path = os.path.join(HERE, "dataparser.npz")
    np.savez_compressed(path, **store)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB, {len(store)} arrays)")


if __name__ == "__main__":
    main()
