"""the NeRF-OSR / synthetic on-disk parsers and the 4-channel mask dataset against golden vectors
obtained by running the reference's own parsers on the same seeded fixture (tests/golden/make_golden_dataparser.py).
File names / index maps / uint8 images / masks: exact.  Poses and intrinsics: 1e-6 (same float32 arithmetic)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import dataset_fixture as fx  # noqa: E402

from neusky_amd.data import dataparsers as dp  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "dataparser.npz"))


def names(paths):
    return np.array([os.path.basename(p) if p is not None else "" for p in paths])


@pytest.fixture(scope="module")
def osr_root(tmp_path_factory):
    return fx.build_nerfosr(str(tmp_path_factory.mktemp("osr")))


@pytest.fixture(scope="module")
def syn_scene(tmp_path_factory):
    return fx.build_synthetic(str(tmp_path_factory.mktemp("syn")))


def check_outputs(prefix, out):
    cam = out.cameras
    np.testing.assert_allclose(cam.camera_to_worlds.numpy(), G[f"{prefix}_c2w"], rtol=0, atol=1e-6)
    for k in ("fx", "fy", "cx", "cy"):
        np.testing.assert_allclose(getattr(cam, k).numpy(), G[f"{prefix}_{k}"], rtol=0, atol=1e-6)
    assert list(names(out.image_filenames)) == list(G[f"{prefix}_images"])
    md = out.metadata
    assert list(names(md["out_of_view_frustum_objects_masks"])) == list(G[f"{prefix}_object_masks"])
    assert list(md["width_height"]) == list(G[f"{prefix}_width_height"])
    if f"{prefix}_session_sizes" in G.files:
        s2i = md["session_to_indices"]
        assert [len(s2i[k]) for k in sorted(s2i)] == list(G[f"{prefix}_session_sizes"])
        assert [i for k in sorted(s2i) for i in s2i[k]] == list(G[f"{prefix}_session_members"])
        assert [md["indices_to_session"][i] for i in range(len(out.image_filenames))] == list(G[f"{prefix}_index_session"])
    else:
        assert md["session_to_indices"] is None
    tk = sorted(md["test_eval_mask_dict"])
    assert tk == list(G[f"{prefix}_eval_mask_idx"])
    assert list(names([md["test_eval_mask_dict"][k] for k in tk])) == list(G[f"{prefix}_eval_mask_names"])
    if f"{prefix}_semantic_files" in G.files:
        assert list(names(md["semantics"].filenames)) == list(G[f"{prefix}_semantic_files"])


def osr_config(root, **over):
    kw = dict(data=root, scene="site1", crop_to_equal_size=True, mask_vegetation=True, session_holdout_indices=[0, 0, 0],
              mask_out_of_view_frustum_objects=True)  # neusky_config.py:47-56
    kw.update(over)
    return dp.NeRFOSRCityScapesDataParserConfig(**kw)


@pytest.mark.parametrize("split", ["train", "val", "test"])
def test_nerfosr_crop_config_matches_reference(osr_root, split):
    parser = osr_config(osr_root).setup()
    out = parser.get_dataparser_outputs(split)
    check_outputs(f"osr_crop_{split}", out)
    ds = dp.NeuSkyDataset(out, split="validation" if split == "val" else split)
    assert len(ds) == len(out.image_filenames) > 0
    for i in range(len(ds)):
        np.testing.assert_array_equal(ds.get_numpy_image(i), G[f"osr_crop_{split}_img{i}"])
        mask = ds.get_mask(i)
        assert mask.shape[-1] == 4 and mask.dtype == torch.float32
        np.testing.assert_array_equal(mask.numpy().astype(np.uint8), G[f"osr_crop_{split}_mask{i}"])
    if split == "train":
        env = out.metadata["envmap_cameras"]
        np.testing.assert_array_equal(env.fx.numpy(), G["osr_env_fx"])
        np.testing.assert_array_equal(env.cx.numpy(), G["osr_env_cx"])
        np.testing.assert_array_equal(env.cy.numpy(), G["osr_env_cy"])
        np.testing.assert_array_equal(env.camera_to_worlds.numpy(), G["osr_env_c2w"])
        assert [os.path.relpath(p, osr_root) for p in out.metadata["envmap_filenames"]] == list(G["osr_env_files"])
        assert ds.get_envmap(0).shape == (3, 8, 16) and float(ds.get_envmap(0).max()) <= 1.0
        assert ds.metadata["num_sessions"] == 3


def test_nerfosr_pad_half_resolution_matches_reference(osr_root):
    cfg = osr_config(osr_root, crop_to_equal_size=False, pad_to_equal_size=True, mask_vegetation=False,
                     include_sidewalk_in_ground_mask=False, mask_out_of_view_frustum_objects=False, orientation_method="up",
                     center_method="poses", scale_factor=0.5)
    out = cfg.setup().get_dataparser_outputs("train")
    check_outputs("osr_pad_train", out)
    ds = dp.NeuSkyDataset(out, scale_factor=0.5, split="train")
    for i in (0, 3):
        np.testing.assert_array_equal(ds.get_numpy_image(i), G[f"osr_pad_train_img{i}"])
        np.testing.assert_array_equal(ds.get_mask(i).numpy().astype(np.uint8), G[f"osr_pad_train_mask{i}"])


def test_nerfosr_error_behaviour(osr_root, tmp_path):
    with pytest.raises(AssertionError):  # :215-218
        osr_config(osr_root, pad_to_equal_size=True).setup()
    with pytest.raises(AssertionError):  # one hold-out index per session on eval splits (:359-363)
        osr_config(osr_root, session_holdout_indices=[0, 0]).setup().get_dataparser_outputs("test")
    with pytest.raises(ValueError):  # the hold-out frame may not carry an eval mask (:437-441)
        osr_config(osr_root, session_holdout_indices=[1, 1, 1]).setup().get_dataparser_outputs("test")
    # a scene without a cityscapes_mask folder: ValueError unless inference is requested, then NotImplementedError (:376-383)
    import shutil
    bare = tmp_path / "bare"
    shutil.copytree(osr_root, bare)
    shutil.rmtree(bare / "lk2" / "final" / "train" / "cityscapes_mask")
    with pytest.raises(ValueError):
        osr_config(str(bare)).setup().get_dataparser_outputs("train")
    with pytest.raises(NotImplementedError):
        osr_config(str(bare), run_segmentation_inference=True).setup().get_dataparser_outputs("train")
    out = osr_config(str(bare), mask_source="original").setup().get_dataparser_outputs("test")
    assert out.metadata["semantics"] is None and len(out.mask_filenames) == 3


@pytest.mark.parametrize("tag,sfm", [("syn", False), ("synsfm", True)])
def test_synthetic_parser_matches_reference(syn_scene, tag, sfm):
    parser = dp.CustomNeuskyDataparserConfig(data=syn_scene, center_method_sfm=sfm).setup()
    for split in ("train", "val", "test"):  # the fixture has no test folder: falls back to the training frames
        out = parser.get_dataparser_outputs(split)
        check_outputs(f"{tag}_{split}", out)
        np.testing.assert_allclose(out.metadata["orientation_rotation"].numpy(), G[f"{tag}_{split}_orientation"], atol=1e-6)
        keys = sorted(k for k in out.metadata if k.startswith("gt_") and k.endswith("_filenames"))
        assert keys == list(G[f"{tag}_{split}_gt_keys"])
        hits = [(-1.0 if e is None else float(e["rotation"])) for e in out.metadata["gt_envmap_info"]]
        np.testing.assert_allclose(hits, G[f"{tag}_{split}_envmap_hits"])
    val = dp.NeuSkyDataset(parser.get_dataparser_outputs("val"), split="validation")
    item = val[0]  # EXR layers are listed but pyexr is absent here: the reader skips them, as the reference does
    assert set(item) >= {"image_idx", "image", "mask"} and item["image"].shape == (18, 24, 3)


def test_pose_normalisation_properties():
    """the restated nerfstudio orientation (parity UNPINNED): 'up' rotates the mean camera-up onto +z, the transform is
    rigid, 'none'/'none' is the identity, and 'focus' puts the point the cameras look at on the origin"""
    g = torch.Generator().manual_seed(0)
    n = 12
    ang = torch.linspace(0, 5.0, n)
    eye = torch.stack([3 * torch.cos(ang), 3 * torch.sin(ang), 0.3 * torch.randn(n, generator=g)], -1)
    tilt = dp.rotation_matrix(torch.tensor([0.0, 0.0, 1.0]), torch.tensor([0.3, -0.2, 1.0]))
    target = torch.tensor([0.5, -0.25, 0.1])
    poses = []
    for e in eye:
        z = (e - target) / torch.linalg.norm(e - target)  # OpenGL: camera looks along -z
        x = torch.linalg.cross(torch.tensor([0.0, 0.0, 1.0]), z)
        x = x / torch.linalg.norm(x)
        y = torch.linalg.cross(z, x)
        poses.append(torch.cat([torch.stack([x, y, z], -1), e[:, None]], -1))
    poses = torch.stack(poses)
    world = torch.cat([tilt, torch.tensor([[1.0], [2.0], [3.0]])], -1)
    poses_w = torch.cat([world[:, :3] @ poses[:, :, :3], world[:, :3] @ poses[:, :, 3:] + world[:, 3:]], -1)
    poses_w = torch.cat([poses_w, torch.tensor([0.0, 0, 0, 1]).expand(n, 1, 4)], 1)
    ident, t0 = dp.auto_orient_and_center_poses(poses_w, "none", "none")
    assert torch.allclose(ident, poses_w[:, :3]) and torch.allclose(t0, torch.eye(4)[:3])
    for method in ("up", "vertical"):
        out, tr = dp.auto_orient_and_center_poses(poses_w, method, "focus")
        R = tr[:, :3]
        assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5) and abs(float(torch.linalg.det(R)) - 1) < 1e-5
        up = out[:, :3, 1].mean(0)
        assert float(up[2] / torch.linalg.norm(up)) > 0.999
        focus_after = R @ (world[:, :3] @ target + world[:, 3]) + tr[:, 3]
        assert float(torch.linalg.norm(focus_after)) < 1e-3
    out, _ = dp.auto_orient_and_center_poses(poses_w, "pca", "poses")
    assert torch.allclose(out[:, :3, 3].mean(0), torch.zeros(3), atol=1e-5)
    with pytest.raises(ValueError):
        dp.auto_orient_and_center_poses(poses_w, "sideways", "poses")


def test_device_datamanager_from_dataset_cpu(osr_root):
    """parsed folders -> image / mask stacks -> the device datamanager (run on CPU tensors here)"""
    from neusky_amd.data.image_datamanager import DeviceImageDataManager
    out = osr_config(osr_root).setup().get_dataparser_outputs("train")
    ds = dp.NeuSkyDataset(out, split="train")
    images, masks = dp.load_stacks(ds)
    assert images.shape == (6, 30, 40, 3) and masks.shape == (6, 30, 40, 4) and masks.dtype == torch.bool
    dm = DeviceImageDataManager.from_dataset(ds, train_num_rays_per_batch=64, device="cpu")
    bundle, batch = dm.next_train(0)
    assert bundle.origins.shape == (64, 3) and batch["image"].shape == (64, 3) and batch["mask"].shape == (64, 4)
    assert bool(batch["mask"][:, 0].all())  # only static pixels are drawn (neusky_pixel_sampler.py:36-46)
    cam = bundle.camera_indices[:, 0]
    np.testing.assert_allclose(bundle.origins.numpy(), out.cameras.camera_to_worlds[cam, :, 3].numpy(), atol=1e-6)
    np.testing.assert_allclose(bundle.pixel_area[:, 0].numpy(), (1.0 / (out.cameras.fx[cam] * out.cameras.fy[cam])).numpy(), rtol=1e-6)
    # unequal frames without crop / pad cannot be stacked
    uneven = dp.NeuSkyDataset(osr_config(osr_root, crop_to_equal_size=False).setup().get_dataparser_outputs("train"))
    with pytest.raises(ValueError):
        dp.load_stacks(uneven)


def test_neusky_datamanager_over_the_parsed_scene(osr_root):
    """the reference's datamanager seam (neusky_datamanager.py:56-288, neusky_config.py:46-64) on the on-disk parser: train / eval datasets,
    resident stacks, the iterator functions the pipeline calls; every batch value is the dataset's own pixel at batch['indices']"""
    from neusky_amd.data.image_datamanager import NeuSkyDataManagerConfig
    cfg = NeuSkyDataManagerConfig(dataparser=osr_config(osr_root), train_num_rays_per_batch=64, eval_num_rays_per_batch=32)
    dm = cfg.setup(device="cpu", test_mode="val", world_size=1, local_rank=0)
    assert len(dm.train_dataset) > 0 and len(dm.eval_dataset) > 0 and dm.num_val == len(dm.eval_dataset)
    assert dm.train_dataset.scene_box["aabb"].shape == (2, 3) and dm.get_param_groups() == {}
    rb, batch = dm.next_train(0)
    assert rb.origins.shape == (64, 3) and rb.directions.shape == (64, 3) and batch["image"].shape == (64, 3) and batch["mask"].shape == (64, 4)
    assert torch.allclose(rb.directions.norm(dim=-1), torch.ones(64), atol=1e-5)
    assert int(rb.camera_indices.min()) >= 0 and int(rb.camera_indices.max()) < len(dm.train_dataset)
    assert bool(batch["mask"][:, 0].all())  # train rays come from the static mask only (neusky_pixel_sampler.py:36-81)
    idx = batch["indices"]
    for r in range(0, 64, 7):  # the batch's values are the dataset's own pixels
        c, y, x = (int(v) for v in idx[r])
        img = torch.from_numpy(dm.train_dataset.get_numpy_image(c)).float() / 255.0
        assert torch.allclose(batch["image"][r], img[y, x, :3], atol=1e-6)
    erb, ebatch = dm.next_eval(0)
    assert erb.origins.shape == (32, 3) and ebatch["image"].shape == (32, 3)
    image_idx, cam_rb, full = dm.next_eval_image(0)
    H, W = full["image"].shape[:2]
    assert cam_rb.origins.shape == (H, W, 3) and cam_rb.camera_indices.shape == (H, W, 1) and full["mask"].shape == (H, W, 4)
    assert 0 <= image_idx and len(dm.eval_dataloader) == len(dm.eval_dataset)
    first = next(iter(dm.eval_dataloader))
    assert first[0].directions.shape == (H, W, 3) and torch.equal(first[1]["image"], full["image"])
    sky = dm.get_sky_ray_bundle(16)
    assert sky.origins.shape == (16, 3)
    hrb, hbatch = dm.get_eval_image_half_bundle("left_image_half", image_index=0, num_rays=8)
    assert hrb.origins.shape == (8, 3) and int(hbatch["indices"][:, 2].max()) < W // 2
