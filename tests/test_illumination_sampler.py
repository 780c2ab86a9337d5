"""IcosahedronSampler (neusky/model_components/illumination_samplers.py:72-123) on the host: the icosphere vertex set is
restated from the published construction of the absent `icosphere` package (parity unpinned), so the checks are the
properties the reference relies on: vertex count 10 nu^2 + 2, unit norm, central symmetry (static upper-hemisphere size
under any rotation), near-uniform spacing, the z > 0 filter and the rotation being a rigid motion."""
import math

import pytest
import torch

from neusky_amd.model_components.illumination import (IcosahedronSamplerConfig, antipodal_sphere, icosphere_vertices,
                                                      random_rotation)


def test_icosphere_vertex_set():
    for nu in (1, 2, 3, 5, 7):
        v = icosphere_vertices(nu).double()
        assert v.shape == (10 * nu * nu + 2, 3)
        assert float((v.norm(dim=1) - 1).abs().max()) < 1e-6
        # centrally symmetric: every -v is a vertex
        assert float(torch.cdist(v, -v).min(1).values.max()) < 1e-6
        # no duplicates, near-uniform: nearest-neighbour distances within a factor 1.5 of each other
        nn = (torch.cdist(v, v) + 10 * torch.eye(v.shape[0], dtype=v.dtype)).min(1).values
        assert float(nn.min()) > 1e-3 and float(nn.max() / nn.min()) < 1.5
    # nu = 1 is the icosahedron itself: edge length 1 / sin(2 pi / 5) for unit circumradius
    v = icosphere_vertices(1).double()
    nn = (torch.cdist(v, v) + 10 * torch.eye(12, dtype=v.dtype)).min(1).values
    assert float((nn - 1.0 / math.sin(2 * math.pi / 5)).abs().max()) < 1e-6


def test_sampler_options():
    s = IcosahedronSamplerConfig(icosphere_order=2, apply_random_rotation=False, remove_lower_hemisphere=True).setup()  # neusky_pipeline.py:157-160
    d = s()
    assert d.shape[1] == 3 and bool((d[:, 2] > 0).all()) and d.shape[0] == 17  # 42 vertices: 17 up, 17 down, 8 on the equator
    dirs, sel = s.on_device("cpu")
    assert sel.tolist() == list(range(17)) and torch.equal(dirs, d)
    s.set_icosphere_order(3)
    assert s.directions.shape[0] < 92 and bool((s.directions[:, 2] > 0).all())
    assert s.icosphere_order_from_num_directions(512) == 51  # the reference's arithmetic, kept as is
    # full sphere + random rotation: rigid motion of the same set, exactly half above the horizon
    full = IcosahedronSamplerConfig(icosphere_order=7, apply_random_rotation=True).setup()
    g = torch.Generator().manual_seed(3)
    R = random_rotation(g)
    d = full(rotation=R)
    assert d.shape == (492, 3)
    assert float((d.double() @ d.double().T - full.directions.double() @ full.directions.double().T).abs().max()) < 1e-5  # Gram matrix kept
    # filter after the rotation (illumination_samplers.py:113-118): data-dependent size on the host path
    cut = IcosahedronSamplerConfig(icosphere_order=4, apply_random_rotation=True, remove_lower_hemisphere=True).setup()
    d = cut(rotation=R)
    assert bool((d[:, 2] > 0).all()) and 70 <= d.shape[0] <= 81


@pytest.mark.gpu
def test_rotated_set_on_the_device_has_exactly_half_above_the_horizon():
    """the rotated set and its upper half from ONE kernel (csrc/samplers.hip: illumination_directions_kernel)"""
    full = IcosahedronSamplerConfig(icosphere_order=7, apply_random_rotation=True).setup()
    R = random_rotation(torch.Generator().manual_seed(3))
    dirs, sel = full.on_device("cuda:0", rotation=R)
    dirs, sel = dirs.cpu(), sel.cpu()
    assert (dirs - full(rotation=R)).abs().max() < 1e-6
    assert sel.numel() == 246 and bool((dirs[sel.long(), 2] > 0).all())
    rest = torch.ones(492, dtype=torch.bool); rest[sel.long()] = False
    assert bool((dirs[rest, 2] <= 0).all())


def test_fixed_icosphere_uses_strict_upper_hemisphere():
    """eval with fix_test_illumination_directions: the unrotated icosphere has vertices ON the equator; the reference's mask is
    the strict z > 0 (neusky_model.py:1650-1657), so they get the constant lower-hemisphere visibility, not a DDF query"""
    s = IcosahedronSamplerConfig(icosphere_order=2, apply_random_rotation=True).setup()
    dirs, sel = s.on_device("cpu", apply_random_rotation=False)
    assert dirs.shape[0] == 42 and sel.numel() == 17 and bool((dirs[sel.long(), 2] > 0).all())
    rest = torch.ones(42, dtype=torch.bool); rest[sel.long()] = False
    assert bool((dirs[rest, 2] <= 0).all()) and int((dirs[rest, 2] == 0).sum()) == 8


def test_default_lattice_unchanged():
    s = IcosahedronSamplerConfig().setup()  # num_directions = 512 (neusky_config.py:97-101): antipodal lattice
    assert torch.equal(s.directions, antipodal_sphere(512))
    dirs, sel = s.on_device("cpu", apply_random_rotation=False)
    assert sel.numel() == 256
