"""Device datamanager against a plain numpy restatement of the pinhole / mask semantics.
Runs on CPU (device='cpu'); the same code path runs on the GPU inside the pipeline."""
import numpy as np
import pytest
import torch

from neusky_amd.data.image_datamanager import DeviceImageDataManager


def _scene(seed=0, N=3, H=12, W=16):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(N, H, W, 3, generator=g)
    u = torch.rand(N, H, W, 4, generator=g)
    masks = torch.stack([u[..., 0] < 0.8, u[..., 1] < 0.6, u[..., 2] < 0.2, u[..., 3] < 0.3], -1)
    q, _ = torch.linalg.qr(torch.randn(N, 3, 3, generator=g))
    c2w = torch.cat([q, torch.randn(N, 3, 1, generator=g) * 0.3], -1)
    return images, masks, c2w


def test_sampling_respects_masks_and_matches_pinhole_math():
    images, masks, c2w = _scene()
    dm = DeviceImageDataManager(images, masks, c2w, fx=20.0, fy=21.0, cx=8.0, cy=6.0, train_num_rays_per_batch=500, device="cpu")
    rb, batch = dm.next_train(0)
    idx = batch["indices"].numpy()
    assert masks.numpy()[idx[:, 0], idx[:, 1], idx[:, 2], 0].all(), "training pixels must have the static channel set"
    np.testing.assert_array_equal(batch["image"].numpy(), images.numpy()[idx[:, 0], idx[:, 1], idx[:, 2]])
    np.testing.assert_array_equal(batch["mask"].numpy(), masks.numpy()[idx[:, 0], idx[:, 1], idx[:, 2]])
    # numpy restatement of the ray maths
    c, y, x = idx[:, 0], idx[:, 1].astype(np.float64), idx[:, 2].astype(np.float64)
    d_cam = np.stack([(x + 0.5 - 8.0) / 20.0, -(y + 0.5 - 6.0) / 21.0, -np.ones_like(x)], -1)
    R = c2w.numpy().astype(np.float64)[c, :, :3]
    d = np.einsum("rij,rj->ri", R, d_cam)
    n = np.linalg.norm(d, axis=-1, keepdims=True)
    np.testing.assert_allclose(rb.directions.numpy(), d / n, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rb.metadata["directions_norm"].numpy(), n, rtol=1e-5)
    np.testing.assert_allclose(rb.origins.numpy(), c2w.numpy()[c, :, 3], rtol=0, atol=0)
    np.testing.assert_array_equal(rb.camera_indices.numpy()[:, 0], c)
    sky = dm.get_sky_ray_bundle(256)
    assert sky.origins.shape == (256, 3) and torch.allclose(sky.directions.norm(dim=-1), torch.ones(256), atol=1e-5)
    # the centre pixel looks along the camera's -z axis
    centre = dm.generate_rays(torch.tensor([[1, 5, 7]]))  # y+0.5 = cy - 0.5 ... not exactly centred: check sign only
    assert float((centre.directions[0] * -c2w[1, :, 2]).sum()) > 0.9


def test_eval_half_bundle_and_uniformity():
    images, masks, c2w = _scene(seed=1, N=2, H=10, W=20)
    dm = DeviceImageDataManager(images, masks, c2w, fx=15.0, fy=15.0, cx=10.0, cy=5.0, train_num_rays_per_batch=4000, device="cpu", num_eval=2)
    rb, batch = dm.get_eval_image_half_bundle("left_image_half", image_index=1, num_rays=300)
    idx = batch["indices"]
    assert (idx[:, 0] == 1).all() and (idx[:, 2] < 10).all() and masks[idx[:, 0], idx[:, 1], idx[:, 2], 0].all()
    # uniform over valid pixels: per-image share of samples ~ share of valid pixels
    _, b = dm.next_train(0)
    share = (b["indices"][:, 0] == 0).float().mean().item()
    want = masks[0, ..., 0].sum().item() / masks[..., 0].sum().item()
    assert abs(share - want) < 0.04


def test_pixel_sets_and_collation_match_the_reference_sampler():
    _check_against_the_reference_sampler("cpu")


@pytest.mark.gpu
def test_pixel_sets_and_collation_match_the_reference_sampler_on_the_gpu():
    """SURVEY 8(f)1 under -m gpu: the same golden comparison with the images, masks, pixel tables and draws resident on cuda:0
    (the device the train step samples on), plus the ray maths of a drawn batch against the numpy pinhole restatement"""
    _check_against_the_reference_sampler("cuda:0")
    images, masks, c2w = _scene()
    dm = DeviceImageDataManager(images, masks, c2w, fx=20.0, fy=21.0, cx=8.0, cy=6.0, train_num_rays_per_batch=500, device="cuda:0")
    rb, batch = dm.next_train(0)
    assert rb.origins.is_cuda and batch["image"].is_cuda and batch["indices"].is_cuda
    idx = batch["indices"].cpu().numpy()
    assert masks.numpy()[idx[:, 0], idx[:, 1], idx[:, 2], 0].all()
    np.testing.assert_array_equal(batch["image"].cpu().numpy(), images.numpy()[idx[:, 0], idx[:, 1], idx[:, 2]])
    np.testing.assert_array_equal(batch["mask"].cpu().numpy(), masks.numpy()[idx[:, 0], idx[:, 1], idx[:, 2]])
    c, y, x = idx[:, 0], idx[:, 1].astype(np.float64), idx[:, 2].astype(np.float64)
    d_cam = np.stack([(x + 0.5 - 8.0) / 20.0, -(y + 0.5 - 6.0) / 21.0, -np.ones_like(x)], -1)
    d = np.einsum("rij,rj->ri", c2w.numpy().astype(np.float64)[c, :, :3], d_cam)
    n = np.linalg.norm(d, axis=-1, keepdims=True)
    np.testing.assert_allclose(rb.directions.cpu().numpy(), d / n, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rb.origins.cpu().numpy(), c2w.numpy()[c, :, 3], rtol=0, atol=0)
    sky = dm.get_sky_ray_bundle(256)
    assert sky.origins.is_cuda and sky.origins.shape == (256, 3)


def _check_against_the_reference_sampler(device):
    """golden from the reference's own NeuSkyPixelSampler (tests/golden/make_golden_sampler.py; nerfstudio's random draw replaced
    by an enumeration of the admissible pixels): the pixels THIS datamanager may draw in each mode are exactly the reference's,
    and a batch is collated the same way (values by [c, y, x], indices[:, 0] remapped through image_idx)"""
    import os
    import numpy as np
    from neusky_amd.data.image_datamanager import DeviceImageDataManager
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixel_sampler.npz"))
    N, H, W = g["image"].shape[:3]
    c2w = torch.eye(4)[:3][None].repeat(N, 1, 1)
    dm = DeviceImageDataManager(torch.from_numpy(g["image"]), torch.from_numpy(g["mask"]), c2w, 10.0, 10.0, W / 2, H / 2, device=device,
                                train_num_rays_per_batch=16, image_idx=torch.from_numpy(g["image_idx"]))
    key = lambda a: sorted(map(tuple, np.asarray(a).tolist()))  # noqa: E731
    remap = lambda p: torch.stack([dm.image_idx[p[:, 0]], p[:, 1], p[:, 2]], 1).cpu().numpy()  # noqa: E731
    assert key(dm.static_pixels.cpu().numpy()) == key(g["train_pixels"])                      # neusky_pixel_sampler.py:36-46
    assert key(remap(dm.sky_pixels)) == key(g["sky_pixels_remapped"])                   # :58-62
    for region in ("left_image_half", "right_image_half", "full_image"):                # :128-146
        assert key(remap(dm.half_pixels(region))) == key(g[f"{region}_pixels_remapped"]), region
    # collation of the very pixels the reference collated (un-remap the first column to stack positions)
    pos = {int(v): i for i, v in enumerate(g["image_idx"].tolist())}
    for name in ("sky", "left_image_half", "full_image"):
        ref_idx = g[f"{name}_batch_indices"]
        stack = torch.from_numpy(np.stack([[pos[int(r[0])] for r in ref_idx], ref_idx[:, 1], ref_idx[:, 2]], 1))
        b = dm.collate(stack.to(device))
        assert str(b["image"].device) == str(torch.device(device))
        assert torch.equal(b["indices"].cpu(), torch.from_numpy(ref_idx)) and torch.equal(b["image"].cpu(), torch.from_numpy(g[f"{name}_batch_image"]))
        assert torch.equal(b["mask"].float().cpu(), torch.from_numpy(g[f"{name}_batch_mask"]))
    # and what the manager actually draws stays inside those sets
    rb, batch = dm.next_train(0)
    assert set(map(tuple, batch["indices"].cpu().tolist())) <= set(key(remap(dm.static_pixels)))
    sky = dm._draw(dm.sky_pixels, 64)
    assert set(map(tuple, remap(sky).tolist())) <= set(key(g["sky_pixels_remapped"]))


@pytest.mark.gpu
def test_pipeline_trains_on_the_parsed_nerfosr_fixture(tmp_path):
    """the method's own datamanager (configs/neusky_config.py: NeuSkyDataManagerConfig over the NeRF-OSR parser, images and masks on the
    device) under the pipeline: two train iterations on the on-disk fixture scene, finite falling-or-equal objective, gradients in every
    optimizer group, and an eval batch / eval image through the same objects"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import dataset_fixture as fx
    from neusky_amd.data import dataparsers as dp
    from neusky_amd.data.image_datamanager import NeuSkyDataManagerConfig
    from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
    from util_step import randomise, small_pipeline_config
    root = fx.build_nerfosr(str(tmp_path))
    cfg = small_pipeline_config(R=64, num_prop=(24, 12), S=8, D=24)
    cfg.datamanager = NeuSkyDataManagerConfig(
        dataparser=dp.NeRFOSRCityScapesDataParserConfig(data=root, scene="site1", crop_to_equal_size=True, mask_vegetation=True,
                                                        session_holdout_indices=[0, 0, 0], mask_out_of_view_frustum_objects=True),
        train_num_rays_per_batch=64, eval_num_rays_per_batch=32)
    torch.manual_seed(0)
    pipe = cfg.setup(device="cuda:0")
    pipe.train()
    randomise(pipe)
    assert int(pipe.num_train_data) == len(pipe.datamanager.train_dataset) > 0
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    losses = [float(train_iteration(pipe, opt, 1000 + i)[0]) for i in range(2)]
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    for g in opt.groups:
        assert float(g.flat_g.abs().max()) > 0.0 or g.name == "visibility_sigmoid", g.name
    rb, batch = pipe.datamanager.next_eval(0)
    assert rb.origins.is_cuda and batch["image"].shape == (32, 3)
    idx, cam_rb, full = pipe.datamanager.next_eval_image(0)
    assert cam_rb.origins.shape[:2] == full["image"].shape[:2]


def _fixture_datamanager(tmp_path, device="cpu", test_mode="val", method="per_image", rays=256):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import dataset_fixture as fx
    from neusky_amd.data import dataparsers as dp
    from neusky_amd.data.image_datamanager import NeuSkyDataManagerConfig
    root = fx.build_nerfosr(str(tmp_path))
    cfg = NeuSkyDataManagerConfig(
        dataparser=dp.NeRFOSRCityScapesDataParserConfig(data=root, scene="site1", crop_to_equal_size=True, mask_vegetation=True,
                                                        session_holdout_indices=[0, 0, 0], mask_out_of_view_frustum_objects=True),
        train_num_rays_per_batch=64, eval_num_rays_per_batch=rays)
    return cfg.setup(device=device, test_mode=test_mode, eval_latent_optimise_method=method)


def test_eval_half_bundle_draws_from_every_eval_image(tmp_path):
    """neusky_datamanager.py:288-305: the eval-latent fit samples the whole cached eval batch -- every eval image's latent row gets
    rays (a default of image 0 left all other eval latents at their zero initialisation); num_val / num_test as :114-119"""
    dm = _fixture_datamanager(tmp_path)
    n_eval = len(dm.eval_dataset)
    assert n_eval > 1 and dm.num_val == n_eval
    assert dm.num_test == len(dm.dataparser.get_dataparser_outputs(split="test").image_filenames)
    seen = set()
    for _ in range(4):
        rb, batch = dm.get_eval_image_half_bundle(sample_region="left_image_half")
        assert (batch["indices"][:, 2] < dm.eval.W // 2).all() and batch["mask"][:, 0].all()
        assert torch.equal(rb.camera_indices[:, 0], batch["indices"][:, 0])
        seen |= set(batch["indices"][:, 0].tolist())
    assert seen == set(range(n_eval)), seen
    rb, batch = dm.get_eval_image_half_bundle(sample_region="full_image", image_index=1)
    assert set(batch["indices"][:, 0].tolist()) == {1}


def test_nerfosr_session_modes(tmp_path):
    """neusky_datamanager.py:120-122,183-233,239-253,307-330: one latent per capture session; the optimise bundle comes from the held-out
    image of each session, the compare bundle from the images with an evaluation mask, image indices are replaced by session indices"""
    dm = _fixture_datamanager(tmp_path, test_mode="test", method="nerf_osr_holdout")
    md = dm.eval_dataset.metadata
    n_sessions = len(md["session_to_indices"])
    assert dm.num_val == dm.num_test == n_sessions == 3
    for stage, images in (("optimise", dm.holdout_indices), ("compare", dm.compare_indices)):
        rb, batch = dm.get_nerfosr_lighting_eval_bundle(stage)
        sess = batch["indices"][:, 0]
        assert set(sess.tolist()) <= {md["indices_to_session"][i] for i in images} and torch.equal(rb.camera_indices[:, 0], sess)
        assert batch["mask"][:, 0].all() and batch["image"].shape == (256, 3)
    idx, cam_rb, full = dm.next_eval_image(0)
    assert idx == md["indices_to_session"][dm.compare_indices[0]] and int(cam_rb.camera_indices.unique()) == idx
    with pytest.raises(ValueError):
        _fixture_datamanager(tmp_path, method="per_image").get_nerfosr_lighting_eval_bundle("optimise")


def test_datamanager_generator_state_round_trip(tmp_path):
    """exact resume (utils/checkpoints.py): the train / eval generators are part of the checkpoint"""
    dm = _fixture_datamanager(tmp_path)
    dm.next_train(0)
    state = dm.state_dict()
    a = dm.next_train(1)[1]["indices"].clone()
    e = dm.next_eval(1)[1]["indices"].clone()
    dm.next_train(2)
    dm.load_state_dict(state)
    assert torch.equal(dm.next_train(1)[1]["indices"], a) and torch.equal(dm.next_eval(1)[1]["indices"], e)
