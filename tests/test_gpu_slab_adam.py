"""`neusky_amd.optimizers.SlabAdam` (what `ns-train neusky` gets for its five Adam groups, neusky_config.py:216-237): torch.optim.Adam's
update as one fused launch per group.  Step by step against torch.optim.Adam(eps=1e-15) on the same gradients -- gathered gradients,
gradients that already sit in the pipeline's slab, a parameter without a gradient, a changing learning rate -- and its state dict
round trip."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 35), (64,), (1, 64), (1,), (7, 3, 5), (4096, 2)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]


def test_slab_adam_follows_torch_adam_step_by_step():
    from neusky_amd.optimizers import SlabAdam
    a, b = _params(0), _params(0)
    ref = torch.optim.Adam(a, lr=1e-2, eps=1e-15)
    opt = SlabAdam(b, lr=1e-2, eps=1e-15)
    assert all(torch.equal(x, y) for x, y in zip(a, b)), "re-homing must not change a parameter"
    g = torch.Generator().manual_seed(1)
    for it in range(6):
        for x, y in zip(a, b):
            gr = (torch.randn(x.shape, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))).to(DEV)
            x.grad, y.grad = gr.clone(), gr.clone()
        for o in (ref, opt):
            o.param_groups[0]["lr"] = 1e-2 * (0.9 ** it)  # a scheduler at work
        ref.step(); opt.step()
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), (it, float((x - y).abs().max()))
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 6.0
    for i, x in enumerate(a):
        for k in ("exp_avg", "exp_avg_sq"):  # (per tensor, against its largest element: a moment is a sum of terms of mixed sign and size)
            r, o_ = ref.state[x][k], sd["state"][i][k]
            assert float((r - o_).abs().max()) <= 2e-6 * float(r.abs().max()), (i, k)
    # state dict round trip into a fresh optimizer: the next step agrees
    c = [torch.nn.Parameter(y.detach().clone()) for y in b]
    opt2 = SlabAdam(c, lr=1e-2, eps=1e-15)
    opt2.load_state_dict(sd)
    for y, z in zip(b, c):
        gr = torch.randn(y.shape, generator=g).to(DEV)
        y.grad, z.grad = gr.clone(), gr.clone()
    opt.step(); opt2.step()
    assert all(torch.equal(y, z) for y, z in zip(b, c))


def test_slab_adam_reads_the_pipeline_slab_in_place_and_skips_nothing():
    """gradients that are views of a GradientSlab (the pipeline's, at world_size > 1 or under graph replay) are stepped from the slab
    range itself; a parameter whose .grad is None (unused in the step: its slot is zero) does not move"""
    from neusky_amd.distributed import GradientSlab
    from neusky_amd.optimizers import SlabAdam
    a, b = _params(3), _params(3)
    slab = GradientSlab({"fields": b})
    opt = SlabAdam(b, lr=1e-3, eps=1e-15)
    ref = torch.optim.Adam(a, lr=1e-3, eps=1e-15)
    g = torch.Generator().manual_seed(4)
    for it in range(3):
        slab.flat.zero_()
        for i, (x, (p, view)) in enumerate(zip(a, slab.views)):
            if i == 3:  # never used
                x.grad, p.grad = None, None
                continue
            gr = torch.randn(x.shape, generator=g).to(DEV)
            x.grad = gr.clone()
            view.copy_(gr)
            p.grad = view
        ref.step(); opt.step()
        rng = opt._slabs[0]["slab_range"]
        assert rng is not None and rng[0].data_ptr() == slab.flat.data_ptr(), "the slab range was not used in place"
        for x, y in zip(a, b):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7)
