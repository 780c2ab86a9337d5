"""ops.zeros: the step's zero-filled temporaries carved from one buffer (one fill per step instead of ~100)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_zero_arena_hands_out_zeroed_independent_regions():
    from neusky_amd import ops
    ops._ARENA.update(buf=None, off=0, need=0, cap=0)
    ops.begin_step(DEV)
    a = ops.zeros(3, 5, device=DEV)          # first step: nothing measured yet -> plain torch.zeros
    assert a.shape == (3, 5) and float(a.abs().sum()) == 0.0
    b = ops.zeros(1000, device=DEV)
    need = ops._ARENA["need"]
    assert need >= 15 + 1000
    ops.begin_step(DEV)                      # second step: one buffer of last step's demand
    buf = ops._ARENA["buf"]
    assert buf is not None and buf.numel() == need
    a2 = ops.zeros(3, 5, device=DEV)
    b2 = ops.zeros(1000, device=DEV)
    c2 = ops.zeros_like(torch.empty(7, 7, device=DEV))  # beyond what was measured: its own allocation
    lo, hi = buf.data_ptr(), buf.data_ptr() + 4 * buf.numel()
    assert lo <= a2.data_ptr() < hi and lo <= b2.data_ptr() < hi and not (lo <= c2.data_ptr() < hi)
    assert a2.data_ptr() % 256 == 0 and b2.data_ptr() % 256 == 0
    assert float(a2.abs().sum()) == 0.0 and float(b2.abs().sum()) == 0.0 and float(c2.abs().sum()) == 0.0
    # regions are independent tensors for autograd: writing one in place does not invalidate a tensor saved from another
    w = torch.ones(1000, device=DEV, requires_grad=True)
    y = (b2 + w) * (b2 + w)                  # saves (b2 + w)
    a2.add_(1.0)                             # in-place write to another region of the same buffer
    y.sum().backward()
    assert torch.allclose(w.grad, torch.full((1000,), 2.0, device=DEV))
    assert not a2._is_view() and a2._version == 1 and b2._version == 0
    # a request larger than the per-region limit never goes to the arena
    big = ops.zeros(ops._ARENA_MAX_REGION + 1, device=DEV)
    assert not (lo <= big.data_ptr() < hi)
    ops._ARENA.update(buf=None, off=0, need=0, cap=0)


def test_zero_arena_off_the_gpu_is_plain_zeros():
    from neusky_amd import ops
    ops._ARENA.update(buf=None, off=0, need=0, cap=0)
    ops.begin_step("cpu")
    z = ops.zeros(4, 4, device="cpu")
    assert ops._ARENA["buf"] is None and float(z.sum()) == 0.0
    ops._ARENA.update(buf=None, off=0, need=0, cap=0)
