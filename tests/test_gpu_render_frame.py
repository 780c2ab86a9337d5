"""Relighting render pass (BASELINE config 5 at test size): chunked, HIP-graph-replayed full-frame forward against the
CPU oracle, chunk-size independence, and z-rotation of the illumination."""
import math

import pytest
import torch

from oracle import neusky_oracle as O
from util_step import oracle_params, oracle_step_cfg, randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def scene():
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=16, D=32, images=4).setup(device=DEV)
    randomise(pipe)
    m = pipe.model
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        m.eval_illumination_latents.copy_((torch.randn(m.eval_illumination_latents.shape, generator=g) * 0.3).to(DEV))
        m.eval_scale.copy_((1 + 0.2 * torch.rand(m.eval_scale.shape, generator=g)).to(DEV))
    pipe.eval()
    H, W = 9, 13
    rb, _ = pipe.datamanager._rays(H * W, torch.Generator().manual_seed(5))
    rb.origins = rb.origins[:1].expand(H * W, 3).contiguous().view(H, W, 3)  # one camera
    rb.directions = rb.directions.view(H, W, 3)
    rb.camera_indices = torch.ones(H, W, 1, dtype=torch.long, device=DEV)
    rb.pixel_area = rb.pixel_area.view(H, W, 1)
    rb.metadata = {"directions_norm": torch.ones(H, W, 1, device=DEV)}
    return pipe, rb, (H, W)


def _oracle(pipe, rb, rotation=None):
    p = oracle_params(pipe)
    m = pipe.model
    light = m.illumination_sampler.directions.double()
    return O.neusky_render({k: v.detach() for k, v in p.items()} | {"field.table": p["field.table"]}, oracle_step_cfg(pipe),
                           rb.origins.reshape(-1, 3).cpu().double(), rb.directions.reshape(-1, 3).cpu().double(),
                           m.eval_illumination_latents[1].detach().cpu().double(), m.eval_scale[1].detach().cpu().double(), light,
                           None if rotation is None else rotation.double())


def test_frame_matches_oracle_and_chunking(scene):
    pipe, rb, (H, W) = scene
    ref = _oracle(pipe, rb)
    full = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=1, chunk=32, use_graph=True)
    assert full["rgb"].shape == (H, W, 3)
    got = full["rgb"].reshape(-1, 3).cpu().double()
    rel = ((got - ref["rgb"]).abs().max() / ref["rgb"].abs().max()).item()
    assert rel < 1e-4, rel  # north-star tolerance on rendered radiance
    assert (full["p2p_dist"].reshape(-1, 1).cpu().double() - ref["p2p_dist"]).abs().max().item() < 1e-4
    assert (full["normal"].reshape(-1, 3).cpu().double() - ref["normal"]).abs().max().item() < 2e-4
    # any chunking (and no graph) gives the same image
    other = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=1, chunk=50, use_graph=False)
    assert (other["rgb"] - full["rgb"]).abs().max().item() < 2e-6


def test_frame_with_z_rotation(scene):
    pipe, rb, _ = scene
    a = 0.9
    rot = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    ref = _oracle(pipe, rb, rot)
    got = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=1, chunk=64, rotation=rot.to(DEV), use_graph=True)
    rel = ((got["rgb"].reshape(-1, 3).cpu().double() - ref["rgb"]).abs().max() / ref["rgb"].abs().max()).item()
    assert rel < 1e-4, rel
