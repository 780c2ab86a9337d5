"""nsky_weight_norm_fwd/bwd (ops.WeightNormFn) against torch's own weight-norm arithmetic + autograd:
W = g v / ||v|| re-ordered and zero padded in one launch each way (the six SDF / colour layers of
neusky/fields/sdf_albedo_field.py:147-161 and the inherited geo net)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(v, g, rows, cols):
    W = g * v / v.norm(dim=1, keepdim=True)
    out = W.new_zeros(len(rows), len(cols))
    for r, sr in enumerate(rows):
        if sr < 0:
            continue
        for c, sc in enumerate(cols):
            if sc >= 0:
                out[r, c] = W[sr, sc]
    return out


@pytest.mark.parametrize("o,i,layout", [(256, 71, "padcols"), (257, 256, "sdf_last"), (256, 295, "colour_in"), (3, 256, "padrows"),
                                         (256, 256, "identity")])
def test_weight_norm_layouts_forward_backward(o, i, layout):
    from neusky_amd import ops
    dev = "cuda:0"
    torch.manual_seed(3)
    v = torch.randn(o, i, device=dev, dtype=torch.float64) * 0.3
    if layout == "padcols":
        v[:, 3:] = 0.0  # geometric init of the first geo layer: only the xyz columns are non-zero
    g = torch.rand(o, 1, device=dev, dtype=torch.float64) + 0.5
    pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
    rows, cols = list(range(o)), list(range(i))
    if layout == "padcols":
        cols = cols + [-1] * (pad4(i) - i)
    elif layout == "sdf_last":
        rows = list(range(1, o)) + [0, -1, -1, -1]
    elif layout == "colour_in":
        cols = list(range(39, i)) + [-1] * 4 + list(range(39)) + [-1]
    elif layout == "padrows":
        rows = rows + [-1] * (pad4(o) - o)
    v64 = v.clone().requires_grad_(True)
    g64 = g.clone().requires_grad_(True)
    ref = _reference(v64, g64, rows, cols)
    probe = torch.randn_like(ref)
    (ref * probe).sum().backward()
    v32 = v.float().requires_grad_(True)
    g32 = g.float().requires_grad_(True)
    maps = ops.weight_norm_maps(o, i, rows, cols, dev)
    out = ops.WeightNormFn.apply(v32, g32, *maps)
    assert out.shape == ref.shape
    assert (out.double() - ref).abs().max().item() < 2e-6 * ref.abs().max().item()
    (out * probe.float()).sum().backward()
    assert (v32.grad.double() - v64.grad).abs().max().item() < 5e-6 * v64.grad.abs().max().item()
    assert (g32.grad.double() - g64.grad).abs().max().item() < 5e-6 * g64.grad.abs().max().item()


def test_weight_norm_maps_reject_missing_rows():
    from neusky_amd import ops
    with pytest.raises(AssertionError):
        ops.weight_norm_maps(4, 4, [0, 1, 2, -1], [0, 1, 2, 3], "cuda:0")
