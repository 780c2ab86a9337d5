"""Eval-latent fitting loop on the device datamanager: the loss goes down, only the eval latents and
scale move, and the decoder / field / DDF parameters are untouched."""
import pytest
import torch

from util_step import randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_fit_latent_codes_for_eval():
    from neusky_amd.data.image_datamanager import DeviceImageDataManager
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, images=4).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    m = pipe.model
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 16, 24
    images = torch.rand(N, H, W, 3, generator=g) * 0.5 + 0.25
    u = torch.rand(N, H, W, 4, generator=g)
    masks = torch.stack([u[..., 0] < 0.95, u[..., 1] < 0.5, u[..., 2] < 0.1, u[..., 1] >= 0.5], -1)  # sky = not fg
    c2w = torch.zeros(N, 3, 4)
    c2w[:, :, :3] = torch.eye(3)
    c2w[:, :, 3] = torch.tensor([[0.1, 0.0, 0.0], [-0.1, 0.1, 0.0]])
    dm = DeviceImageDataManager(images, masks, c2w, fx=30.0, fy=30.0, cx=W / 2, cy=H / 2, train_num_rays_per_batch=64, device=DEV, num_eval=2)
    before = {n: p.detach().clone() for n, p in pipe.named_parameters()}
    trace = m.fit_latent_codes_for_eval(dm, global_step=0, steps=60, log_every=1)
    losses = torch.stack(trace).cpu()
    assert torch.isfinite(losses).all()
    assert losses[-10:].mean() < losses[:5].mean(), (losses[:5], losses[-10:])
    moved = [n for n, p in pipe.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert sorted(moved) == ["_model.eval_illumination_latents", "_model.eval_scale"], moved
    assert not m.fitting_eval_latents and m.field.glin0.weight_v.requires_grad and m.field.encoding.params.requires_grad
