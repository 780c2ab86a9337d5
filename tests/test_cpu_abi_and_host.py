"""CPU-only checks: the C-ABI library loads and exports every symbol include/neusky_hip.h declares
(no compute calls without a GPU), and the host-side logic (geometry, schedulers, configs, samplers)."""
import ctypes
import math
import os
import re

import numpy as np
import torch

from oracle import neusky_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "neusky_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(nsky_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 15
    lib = ctypes.CDLL(os.path.join(ROOT, "neusky_amd", "libneusky_hip.so"))
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    lib.nsky_abi_version.restype = ctypes.c_int
    assert lib.nsky_abi_version() == 16
    lib.nsky_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.nsky_last_error(), bytes)


def test_binding_argument_errors_without_gpu():
    from neusky_amd import hip
    d = hip.GemmDesc()  # all-null descriptor must be rejected before any launch
    rc = hip._gemm(ctypes.byref(d), None)
    assert rc < 0 and b"null" in hip._lib.nsky_last_error()


def test_hash_geometry_matches_oracle_and_survey():
    from neusky_amd.encoding import HashGridGeometry
    for kw in [dict(smoothstep=True), dict(n_levels=5, log2_hashmap_size=17, max_res=64), dict(n_levels=5, log2_hashmap_size=17, max_res=256)]:
        g = HashGridGeometry(**kw)
        okw = {k: v for k, v in kw.items()}
        c = O.HashGridCfg(**okw)
        assert g.offsets == c.offsets and g.resolutions == c.resolutions and np.allclose(g.scales, c.scales)
    g = HashGridGeometry()
    assert g.n_params == 6098120  # 12.2 M parameters
    assert g.resolutions[0] == 16 and g.resolutions[-1] == 2048


def test_schedulers_and_optimizer_groups():
    from neusky_amd.engine import CosineDecaySchedulerConfig, ExponentialDecaySchedulerConfig, neusky_optimizers
    c = CosineDecaySchedulerConfig()
    assert c.factor(0) == 0.0 and abs(c.factor(500) - 1.0) < 1e-12 and abs(c.factor(100001) - 0.05) < 1e-9
    e = ExponentialDecaySchedulerConfig(lr_final=1e-5, lr_init=1e-2)
    assert abs(e.factor(0) - 1.0) < 1e-12 and abs(e.factor(100001) - 1e-3) < 1e-9
    w = ExponentialDecaySchedulerConfig(lr_final=1e-4, warmup_steps=4000, lr_init=1e-3)
    assert w.factor(0) < 1e-4 and abs(w.factor(4000) - 1.0) < 1e-9
    assert sorted(neusky_optimizers()) == ["ddf_field", "fields", "illumination_field", "proposal_networks", "visibility_sigmoid"]


def test_method_spec_and_config_defaults():
    from neusky_amd.configs.neusky_config import NeuSky
    cfg = NeuSky.config
    assert cfg.method_name == "neusky" and not cfg.mixed_precision
    m = cfg.pipeline.model
    assert m.loss_coefficients["hashgrid_density_loss"] == 1e-4 and m.loss_inclusions["visibility_sigmoid_loss"]["target_max_scale"] == 25
    assert m.sdf_to_visibility_stop_gradients == "depth" and m.only_upperhemisphere_visibility
    d = cfg.pipeline.visibility_field.ddf_field
    assert (d.hidden_layers, d.hidden_features, d.mapping_layers, d.mapping_features) == (5, 256, 5, 256)
    assert cfg.pipeline.datamanager.train_num_rays_per_batch == 1024


def test_vmf_sampler_statistics():
    """G10: mean direction and concentration of the vMF sampler (RNG streams are not portable)."""
    from neusky_amd.model_components.ddf_sampler import VMFDDFSampler, VMFDDFSamplerConfig
    s = VMFDDFSampler(VMFDDFSamplerConfig(num_samples_on_sphere=4, num_rays_per_sample=4000, concentration=20.0))
    g = torch.Generator().manual_seed(0)
    rb = s.generate_ddf_samples(4, 4000, generator=g)
    o, d = rb.origins.view(4, 4000, 3), rb.directions.view(4, 4000, 3)
    assert torch.allclose(o.norm(dim=-1), torch.ones(4, 4000), atol=1e-5) and (o[..., 2] >= 0).all()
    mean_cos = (d * -o).sum(-1).mean(1)
    kappa = 20.0
    ideal = 1 / math.tanh(kappa) - 1 / kappa  # E[cos] of an exact vMF on S^2 = 0.95
    # the reference accepts on `test >= -e` (ddf_sampler.py:220) instead of `test >= log(u)`: its draw is the
    # Wood proposal truncated, slightly less concentrated than vMF(20).  Both restatements reproduce that.
    assert ((mean_cos > 0.92) & (mean_cos < ideal)).all(), mean_cos
    P, D = O.vmf_ddf_rays(4, 4000, 20.0, 1.0, torch.Generator().manual_seed(1))
    mc = (D.view(4, 4000, 3) * -P.view(4, 4000, 3)).sum(-1).mean(1)
    assert (mc.float().mean() - mean_cos.mean()).abs() < 0.004, (mc, mean_cos)


def test_illumination_sampler_and_invariance():
    from neusky_amd.model_components.illumination import IcosahedronSampler, IcosahedronSamplerConfig, RENIField, random_rotation
    s = IcosahedronSampler(IcosahedronSamplerConfig(num_directions=512))
    d = s(apply_random_rotation=False)
    assert d.shape == (512, 3) and torch.allclose(d.norm(dim=-1), torch.ones(512), atol=1e-5)
    assert abs(int((d[:, 2] > 0).sum()) - 256) <= 1  # Dv ~ D/2
    r = random_rotation(torch.Generator().manual_seed(3))
    assert torch.allclose(r @ r.T, torch.eye(3), atol=1e-5) and abs(float(torch.det(r)) - 1) < 1e-5
    # SO(2)-about-z invariance of the decoder inputs (the property RENI++ is built on)
    g = torch.Generator().manual_seed(4)
    Z, dd = torch.randn(7, 9, 3, generator=g), torch.nn.functional.normalize(torch.randn(7, 3, generator=g), dim=-1)
    a = 0.7
    Rz = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1.0]])
    c0, x0 = RENIField.invariant_inputs(Z, dd)
    c1, x1 = RENIField.invariant_inputs(Z @ Rz.T, dd @ Rz.T)
    assert torch.allclose(c0, c1, atol=1e-5) and torch.allclose(x0, x1, atol=1e-5)
    oc, ox = O.reni_invariant_inputs(Z.double(), dd.double())
    assert torch.allclose(oc.float(), c0, atol=1e-6) and torch.allclose(ox.float(), x0, atol=1e-6)


def test_oracle_properties_of_unpinned_pieces():
    """self-consistency of the externally-defined (unpinned) restatements"""
    g = torch.Generator().manual_seed(5)
    a = torch.rand(6, 9, 1, generator=g, dtype=torch.float64) * 0.5
    w, T = O.weights_from_alphas(a)
    assert (w.sum(1) <= 1 + 1e-6).all() and torch.allclose(w.sum(1) + T[:, -1], torch.ones(6, 1, dtype=torch.float64), atol=1e-5)
    # pdf sampler: sorted bins inside the parent span, indices in range
    bins = torch.sort(torch.rand(6, 11, generator=g, dtype=torch.float64), -1).values
    nb, inds = O.pdf_sample_bins(bins, torch.rand(6, 10, generator=g, dtype=torch.float64), 7, torch.rand(6, 1, generator=g, dtype=torch.float64))
    assert (nb[:, 1:] >= nb[:, :-1]).all() and (nb >= bins[:, :1]).all() and (nb <= bins[:, -1:]).all()
    assert inds.min() >= 0 and inds.max() <= 11
    # contraction maps everything into the radius-2 ball and is the identity inside the unit ball
    x = torch.randn(100, 3, generator=g, dtype=torch.float64) * 3
    c = O.scene_contraction(x)
    assert (c.abs().max(-1).values < 2).all()
    small = x / (x.abs().max(-1, keepdim=True).values * 1.5)
    assert torch.equal(O.scene_contraction(small), small)
    # hash grid determinism + finite-difference gradient w.r.t. position (smoothstep is C1)
    cfg = O.HashGridCfg(n_levels=4, log2_hashmap_size=10, max_res=64, smoothstep=True)
    tab = torch.randn(cfg.n_params, 2, generator=g, dtype=torch.float64)
    p = torch.rand(20, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    f = O.hash_grid_encode(p, tab, cfg)
    assert torch.equal(f, O.hash_grid_encode(p, tab, cfg))
    gr = torch.autograd.grad(f[:, 3].sum(), p)[0]
    eps = 1e-6
    pp = p.detach().clone(); pp[:, 1] += eps
    fd = (O.hash_grid_encode(pp, tab, cfg)[:, 3] - f[:, 3].detach()) / eps
    assert torch.allclose(fd, gr[:, 1], atol=1e-4, rtol=1e-3)


def test_bench_starts_its_own_ranks_cpu_selftest():
    """bench.py's launch path without a GPU (`--launcher-selftest`): the parent starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT, the ranks rendezvous over gloo and all-reduce a known answer, ONLY rank 0's one line reaches the
    parent's stdout (library banners and the other ranks' output go to stderr), and a failing rank gives a non-zero exit and no line"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launcher-selftest", "ok"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d == {"selftest": "ok", "ranks": 3, "sum": 6.0, "launcher": "self-spawned"}
    assert "a line of rank 1" in out.stderr and "a line of rank 2" in out.stderr
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest", "fail"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode == 7 and bad.stdout.strip() == "" and "rank 1 exited with code 7" in bad.stderr
