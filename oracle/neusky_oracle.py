"""CPU ORACLE for the NeuSky hot path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module, and only as the checker / the timed CPU baseline.  The product (`neusky_amd/`) never
imports it and has no CPU fallback.

It is a plain torch-CPU restatement (dtype-parametric: float64 for checking, float32 for the
timed baseline) of the algorithm of JADGardner/neusky's per-ray train/render step.  Every
function cites the reference file:line it follows (paths relative to /root/reference).

PARITY STATUS
  * pinned   - functions restating IN-TREE reference code are checked against golden vectors
               produced by importing the reference itself (tests/golden/make_golden.py, G1-G11):
               linear_to_srgb, ray_sphere_*, lambertian_render, compute_visibility, local frame,
               DDF head / DDF model plumbing, losses, FiLM-SIREN (neusky/utils/siren.py),
               sample_illumination index plumbing, field output plumbing.
  * UNPINNED - "parity unpinned": functions restating EXTERNAL, un-vendored, un-pinned dependencies
               whose source is absent from /root/reference: tiny-cuda-nn
               HashGrid (HEAD, docker/Dockerfile:73-78), nerfstudio (mainline, docker/Dockerfile:89-93:
               SDFField.forward_geonetwork/get_alpha/LearnedVariance, NeRFEncoding, SceneContraction,
               ProposalNetworkSampler/UniformSampler/PDFSampler, HashMLPDensityField, renderers,
               interlevel_loss, monosdf_normal_loss), ns_reni (empty submodule: RENIField).  These
               follow the published algorithms as recalled in SURVEY.md Appendix A and are frozen
               here as THIS project's definition; they are exercised by property tests only.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# =====================================================================================
# small in-tree utilities (PINNED by G1/G2)
# =====================================================================================
def linear_to_srgb(color: Tensor) -> Tensor:
    """neusky/utils/utils.py:11-31 (use_quantile=False)."""
    color = torch.where(color <= 0.0031308, 12.92 * color, 1.055 * torch.pow(torch.abs(color), 1 / 2.4) - 0.055)
    return torch.clamp(color, 0.0, 1.0)


def ray_sphere_intersection_free(positions: Tensor, directions: Tensor, radius: float) -> Tensor:
    """neusky/utils/utils.py:68-93 - far root, directions assumed unit, no clamp."""
    b = 2 * (directions * positions).sum(-1)
    c = (positions * positions).sum(-1) - radius**2
    disc = b**2 - 4 * c
    t = torch.max((-b - torch.sqrt(disc)) / 2, (-b + torch.sqrt(disc)) / 2)
    return positions + t[..., None] * directions


def ray_sphere_intersection_clamped(positions: Tensor, directions: Tensor, radius: float) -> Tensor:
    """neusky/models/neusky_model.py:1590-1622 - normalises directions, clamps discriminant >= 0."""
    directions = directions / torch.norm(directions, dim=-1, keepdim=True)
    b = 2 * (directions * positions).sum(-1)
    c = (positions * positions).sum(-1) - radius**2
    disc = torch.clamp(b**2 - 4 * c, min=0.0)
    t = torch.max((-b - torch.sqrt(disc)) / 2, (-b + torch.sqrt(disc)) / 2)
    return positions + t[..., None] * directions


# =====================================================================================
# A1 / A11  hemisphere integral + alpha composite (PINNED by G3)
# =====================================================================================
def lambertian_render(albedo: Tensor, normals: Tensor, dirs: Tensor, cam_colours: Tensor, cam_of_ray: Tensor,
                      vis: Optional[Tensor], bg: Tensor, weights: Tensor, training: bool = True) -> Tensor:
    """neusky/model_components/renderers.py:60-130 (+ eval clamp :173-174), on COMPACT inputs.

    The reference receives light_directions/light_colors/visibility broadcast to [R*S, D, *]
    (neusky_model.py:512-525, 1755-1759); every sample shares the D directions, every sample of a
    ray shares that ray's camera colours and visibility row, so the same arithmetic is written on
    albedo/normals [R,S,3], dirs [D,3], cam_colours [U,D,3] + cam_of_ray [R], vis [R,D], bg [R,3],
    weights [R,S]."""
    dot = torch.einsum("rsi,ji->rsj", normals, dirs).clamp(0.0, 1.0)  # :93-98
    count = (dot > 0).to(dot.dtype).sum(-1, keepdim=True)  # :101
    count = torch.where(count > 0, count, torch.ones_like(count))  # :104
    dot = dot / count  # :106
    if vis is not None:
        dot = dot * vis[:, None, :]  # :108-110
    cols = cam_colours[cam_of_ray]  # [R,D,3]
    radiance = albedo * torch.einsum("rsj,rjc->rsc", dot, cols)  # :113
    comp = (weights[..., None] * radiance).sum(-2)  # :122
    acc = weights.sum(-1, keepdim=True)  # :123
    comp = comp + bg * (1.0 - acc)  # :127
    rgb = linear_to_srgb(comp)  # :128
    if not training:
        rgb = rgb.clamp(0.0, 1.0)  # :173-174
    return rgb


def composite_aux(weights: Tensor, starts: Tensor, ends: Tensor, normals: Tensor, albedo: Tensor):
    """neusky_model.py:591-595, 812-813 via nerfstudio renderers [UNPINNED external]:
    DepthRenderer(method='expected'): sum(w*mid)/(sum(w)+1e-10) clipped to [min,max] of mids;
    AccumulationRenderer: sum(w); SemanticRenderer-style normal: sum(w*n);
    RGBRenderer(background white): sum(w*a) + 1*(1-sum(w))."""
    steps = (starts + ends) / 2
    acc = weights.sum(-1, keepdim=True)
    depth = (weights * steps).sum(-1, keepdim=True) / (acc + 1e-10)
    depth = torch.clip(depth, steps.min(), steps.max())
    normal = (weights[..., None] * normals).sum(-2)
    alb = (weights[..., None] * albedo).sum(-2) + (1.0 - acc)
    return depth, acc, normal, alb


# =====================================================================================
# A7  DDF local frame (PINNED by G5)
# =====================================================================================
def local_frame(positions: Tensor) -> Tensor:
    """neusky/models/ddf_model.py:158-181 -> rotation matrices [M,3,3] with columns (x,y,z)_local."""
    up = torch.tensor([0.0, 0.0, 1.0], dtype=positions.dtype).expand_as(positions)
    y = -positions
    x = torch.linalg.cross(up, y, dim=-1)
    x = x / x.norm(dim=-1, keepdim=True)
    z = torch.linalg.cross(y, x, dim=-1)
    z = z / z.norm(dim=-1, keepdim=True)
    return torch.stack((x, y, z), dim=-1)


def ddf_local_direction(positions: Tensor, directions: Tensor) -> Tensor:
    """neusky/models/ddf_model.py:196-200: einsum('ijl,ij->il', R_loc, d)."""
    return torch.einsum("ijl,ij->il", local_frame(positions), directions)


# =====================================================================================
# external encodings [UNPINNED]
# =====================================================================================
def nerf_encoding(x: Tensor, num_freq: int, min_exp: float, max_exp: float, include_input: bool) -> Tensor:
    """nerfstudio NeRFEncoding: 2*pi*x*2^linspace(min,max,n); sin(cat(xf, xf+pi/2))."""
    freqs = 2.0 ** torch.linspace(min_exp, max_exp, num_freq, dtype=x.dtype)
    xs = (2.0 * math.pi * x[..., None] * freqs).reshape(*x.shape[:-1], -1)
    enc = torch.sin(torch.cat([xs, xs + math.pi / 2.0], -1))
    return torch.cat([x, enc], -1) if include_input else enc


def scene_contraction(x: Tensor, order: float = float("inf")) -> Tensor:
    """nerfstudio SceneContraction: x if |x|<1 else (2-1/|x|) x/|x|."""
    mag = torch.linalg.norm(x, ord=order, dim=-1)[..., None]
    safe = torch.where(mag < 1, torch.ones_like(mag), mag)
    return torch.where(mag < 1, x, (2 - (1 / safe)) * (x / safe))


@dataclass
class HashGridCfg:
    """tiny-cuda-nn HashGrid geometry (call sites sdf_albedo_field.py:115-130,
    directional_distance_field.py:139-156)."""
    n_levels: int = 16
    n_features: int = 2
    log2_hashmap_size: int = 19
    base_res: int = 16
    max_res: int = 2048
    smoothstep: bool = False
    scales: List[float] = field(default_factory=list)
    resolutions: List[int] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)

    def __post_init__(self):
        growth = np.exp((np.log(self.max_res) - np.log(self.base_res)) / (self.n_levels - 1)) if self.n_levels > 1 else 1.0
        log2_growth = np.float32(np.log2(np.float32(growth)))
        off = 0
        self.scales, self.resolutions, self.offsets = [], [], []
        for lvl in range(self.n_levels):
            scale = np.float32(np.exp2(np.float32(lvl) * log2_growth) * np.float32(self.base_res) - np.float32(1.0))
            res = int(np.ceil(scale)) + 1
            n = min(res**3, 2**31 - 1)
            n = (n + 7) // 8 * 8
            n = min(n, 1 << self.log2_hashmap_size)
            self.scales.append(float(scale)); self.resolutions.append(res); self.offsets.append(off)
            off += n
        self.offsets.append(off)

    @property
    def n_params(self) -> int:
        return self.offsets[-1]

    @property
    def out_dim(self) -> int:
        return self.n_levels * self.n_features


_PRIMES = (1, 2654435761, 805459861)


def hash_grid_indices(x: Tensor, cfg: HashGridCfg):
    """Corner indices [P,L,8] (int64, absolute rows into the table) and the integer cell origin
    floor(pos) [P,L,3] (as float64).  `pos = fmaf(scale, x, 0.5f)` is evaluated with float32
    rounding (as tcnn does) whatever the working dtype, so the CELL CHOICE is identical to the
    fp32 GPU path; uint32 wrap-around semantics of tcnn (`(uint32_t)(int)floorf(pos)`, dense
    index / coherent prime hash, `% size`) are reproduced with int64 arithmetic masked to 32 bits.
    Integer output: compared BIT-EXACT."""
    P = x.shape[0]
    M32 = 0xFFFFFFFF
    x32 = x.detach().to(torch.float32).to(torch.float64)
    idx_all, fl_all = [], []
    for lvl in range(cfg.n_levels):
        scale = float(np.float32(cfg.scales[lvl]))
        res = cfg.resolutions[lvl]
        size = cfg.offsets[lvl + 1] - cfg.offsets[lvl]
        pos = (x32 * scale + 0.5).to(torch.float32)  # single rounding == fmaf up to double-rounding ties
        fl = torch.floor(pos)
        fl_all.append(fl.to(torch.float64))
        g = fl.to(torch.int64) & M32  # (uint32_t)(int)
        corners = []
        for c in range(8):
            pg = [(g[:, d] + ((c >> d) & 1)) & M32 for d in range(3)]
            # dense index with tcnn's `stride <= hashmap_size` loop condition
            stride, index = 1, torch.zeros(P, dtype=torch.int64)
            for d in range(3):
                if stride <= size:
                    index = (index + pg[d] * stride) & M32
                    stride *= res
            if size < stride:  # hashed level
                index = torch.zeros(P, dtype=torch.int64)
                for d in range(3):
                    index = index ^ ((pg[d] * _PRIMES[d]) & M32)
            corners.append(index % size + cfg.offsets[lvl])
        idx_all.append(torch.stack(corners, -1))
    return torch.stack(idx_all, 1), torch.stack(fl_all, 1)


def hash_grid_encode(x: Tensor, table: Tensor, cfg: HashGridCfg) -> Tensor:
    """tcnn HashGrid forward: x [P,3] (fed RAW, caller normalises) -> features [P, L*F].
    Differentiable w.r.t. table and x (torch autograd), so the reference's
    autograd.grad(create_graph=True) flow (sdf_albedo_field.py:231-238) can be followed literally."""
    idx, fl = hash_grid_indices(x, cfg)
    P = x.shape[0]
    vals = table[idx.reshape(-1)].view(P, cfg.n_levels, 8, table.shape[1])  # one gather (one dense grad) per call
    outs = []
    for lvl in range(cfg.n_levels):
        t = x * float(np.float32(cfg.scales[lvl])) + 0.5 - fl[:, lvl].to(x.dtype)
        w = t * t * (3.0 - 2.0 * t) if cfg.smoothstep else t
        acc = 0.0
        for c in range(8):
            wc = 1.0
            for d in range(3):
                wc = wc * (w[:, d] if (c >> d) & 1 else (1.0 - w[:, d]))
            acc = acc + wc[:, None] * vals[:, lvl, c]
        outs.append(acc)
    return torch.cat(outs, -1)


# =====================================================================================
# A2/A4  SDF + albedo field
# =====================================================================================
def weight_norm(v: Tensor, g: Tensor) -> Tensor:
    """torch.nn.utils.weight_norm(dim=0): w = g * v / ||v||_row  (sdf_albedo_field.py:159-160)."""
    return g * v / v.norm(dim=1, keepdim=True)


def softplus100(x: Tensor) -> Tensor:
    return F.softplus(x, beta=100)  # sdf_albedo_field.py:163 (threshold 20 default)


def geo_network(x: Tensor, p: Dict[str, Tensor], cfg: HashGridCfg, contraction_order: float = float("inf")) -> Tensor:
    """nerfstudio SDFField.forward_geonetwork [UNPINNED] (called sdf_albedo_field.py:172,180,233):
    feat = hash((contract(x)+2)/4); h = cat(x, PE6(x), feat); Linear -> Softplus(100) except last."""
    pos = (scene_contraction(x, contraction_order) + 2.0) / 4.0
    feat = hash_grid_encode(pos, p["field.table"], cfg)
    h = torch.cat([x, nerf_encoding(x, 6, 0.0, 5.0, False), feat], -1)
    n = 0
    while f"field.glin{n}.v" in p:
        n += 1
    for l in range(n):
        w = weight_norm(p[f"field.glin{l}.v"], p[f"field.glin{l}.g"])
        h = F.linear(h, w, p[f"field.glin{l}.b"])
        if l < n - 1:
            h = softplus100(h)
    return h


def colour_network(x: Tensor, geo_feat: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """neusky/fields/sdf_albedo_field.py:185-209 (PINNED by G11 with a stand-in geo net)."""
    h = torch.cat([x, nerf_encoding(x, 6, 0.0, 5.0, False), geo_feat], -1)
    n = 0
    while f"field.clin{n}.v" in p:
        n += 1
    for l in range(n):
        w = weight_norm(p[f"field.clin{l}.v"], p[f"field.clin{l}.g"])
        h = F.linear(h, w, p[f"field.clin{l}.b"])
        if l < n - 1:
            h = F.relu(h)
    return torch.sigmoid(h)


def neus_alpha(sdf: Tensor, gradients: Tensor, directions: Tensor, deltas: Tensor, variance: Tensor,
               cos_anneal_ratio: float = 1.0) -> Tensor:
    """nerfstudio SDFField.get_alpha [UNPINNED] (called sdf_albedo_field.py:266)."""
    inv_s = torch.exp(variance * 10.0).clip(1e-6, 1e6)
    true_cos = (directions * gradients).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    nxt = sdf + iter_cos * deltas * 0.5
    prv = sdf - iter_cos * deltas * 0.5
    prev_cdf = torch.sigmoid(prv * inv_s)
    next_cdf = torch.sigmoid(nxt * inv_s)
    return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


def weights_from_alphas(alphas: Tensor) -> Tuple[Tensor, Tensor]:
    """nerfstudio RaySamples.get_weights_and_transmittance_from_alphas [UNPINNED] (App. A.5;
    called neusky_model.py:565-568): alphas [R,S,1] -> weights [R,S,1], T [R,S+1,1]."""
    T = torch.cumprod(torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas + 1e-7], 1), 1)
    return alphas * T[:, :-1], T


def sdf_field_outputs(origins: Tensor, directions: Tensor, starts: Tensor, ends: Tensor, p: Dict[str, Tensor],
                      cfg: HashGridCfg, create_graph: bool = True) -> Dict[str, Tensor]:
    """neusky/fields/sdf_albedo_field.py:211-269: positions = o + d*starts (get_start_positions, :225),
    sdf/feat from the geo net (:233-234), gradients by autograd.grad(create_graph=True) (:235-238),
    albedo (:241), normals (:251), NeuS alpha (:266).  Shapes: origins/directions [R,S,3]; starts/ends [R,S,1]."""
    R, S, _ = origins.shape
    x = (origins + directions * starts).reshape(-1, 3).detach().requires_grad_(True)
    h = geo_network(x, p, cfg)
    sdf, feat = h[:, :1], h[:, 1:]
    grads = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=create_graph, retain_graph=True)[0]
    albedo = colour_network(x, feat, p)
    out = {
        "albedo": albedo.view(R, S, 3), "sdf": sdf.view(R, S, 1), "gradients": grads.view(R, S, 3),
        "normals": F.normalize(grads.view(R, S, 3), p=2, dim=-1),
    }
    out["alpha"] = neus_alpha(out["sdf"], out["gradients"], directions, ends - starts, p["field.variance"])
    return out


def sdf_at_positions(x: Tensor, p: Dict[str, Tensor], cfg: HashGridCfg) -> Tensor:
    """neusky/fields/sdf_albedo_field.py:169-174."""
    return geo_network(x.reshape(-1, 3), p, cfg)[:, :1]


# =====================================================================================
# A9 / A3 / A7  DDF: FiLM-SIREN (PINNED by G8) + field head (PINNED by G6)
# =====================================================================================
def film_siren(x: Tensor, cond: Tensor, p: Dict[str, Tensor], prefix: str = "ddf.") -> Tensor:
    """neusky/utils/siren.py:108-208: mapping = (Linear, LeakyReLU(0.2))*n + Linear (:114-119);
    split first half -> frequencies, second half -> phase shifts (:129-130); freq = raw*15+30 (:200);
    layer: sin(freq*(Wx+b)+phase) (:141-144); final Linear (:207)."""
    h = cond
    n = 0
    while f"{prefix}map_w{n}" in p:
        h = F.leaky_relu(F.linear(h, p[f"{prefix}map_w{n}"], p[f"{prefix}map_b{n}"]), 0.2)
        n += 1
    fo = F.linear(h, p[f"{prefix}map_wo"], p[f"{prefix}map_bo"])
    half = fo.shape[-1] // 2
    freq, phase = fo[..., :half] * 15 + 30, fo[..., half:]
    i = 0
    while f"{prefix}film_w{i}" in p:
        H = p[f"{prefix}film_w{i}"].shape[0]
        z = F.linear(x, p[f"{prefix}film_w{i}"], p[f"{prefix}film_b{i}"])
        x = torch.sin(freq[..., i * H:(i + 1) * H] * z + phase[..., i * H:(i + 1) * H])
        i += 1
    return F.linear(x, p[f"{prefix}out_w"], p[f"{prefix}out_b"])


def ddf_field(positions: Tensor, local_dirs: Tensor, p: Dict[str, Tensor], cfg: HashGridCfg, radius: float) -> Tensor:
    """neusky/fields/directional_distance_field.py:261-306, FiLM + ddf branch:
    cond = cat(p, hash(p)) (:267-268, raw sphere coords fed to tcnn), x = cat(d, NeRF2(d)) (:270-271),
    t = sigmoid(net[...,0]) * 2r (:297-299)."""
    cond = torch.cat([positions, hash_grid_encode(positions, p["ddf.table"], cfg)], -1)
    x = torch.cat([local_dirs, nerf_encoding(local_dirs, 2, 0.0, 2.0, False)], -1)
    return torch.sigmoid(film_siren(x, cond, p)[..., 0]) * (2 * radius)


def ddf_query(positions: Tensor, directions: Tensor, p: Dict[str, Tensor], cfg: HashGridCfg, radius: float) -> Tensor:
    """neusky/models/ddf_model.py:193-219: world rays on the sphere -> expected termination distance."""
    return ddf_field(positions, ddf_local_direction(positions, directions), p, cfg, radius)


# =====================================================================================
# A5  compute_visibility (PINNED by G4)
# =====================================================================================
def compute_visibility(origins: Tensor, ray_dirs: Tensor, depth: Tensor, dirs: Tensor, threshold, scale, radius: float,
                       ddf_fn, only_upper: bool = True, lower_vis: bool = True) -> Dict[str, Tensor]:
    """neusky/models/neusky_model.py:1624-1778 on compact inputs: origins/ray_dirs [R,3] (= [:,0] of the
    samples, :1667-1668), depth [R,1], dirs [D,3] (= illumination_directions[0], :1648).
    ddf_fn(sphere_points [M,3], directions [M,3]) -> dict(expected_termination_dist [M], ...).
    Returns visibility [R,D] (the reference repeats it over S, :1755-1759)."""
    R, D = origins.shape[0], dirs.shape[0]
    if only_upper:
        mask = dirs[:, 2] > 0  # :1653-1657
        sel = dirs[mask]
    else:
        mask = torch.ones(D, dtype=torch.bool)
        sel = dirs
    Dv = sel.shape[0]
    positions = origins + ray_dirs * depth  # :1671
    inside = positions.norm(dim=-1) < radius  # :1674
    fix = ray_sphere_intersection_clamped(origins, ray_dirs, radius) * 0.01 * -ray_dirs  # :1679-1683 (sic: product)
    positions = torch.where(inside[:, None], positions, fix)
    pos = positions[:, None, :].expand(R, Dv, 3).reshape(-1, 3)  # :1685-1690
    dd = sel[None].expand(R, Dv, 3).reshape(-1, 3)
    sphere_pts = ray_sphere_intersection_clamped(pos, dd, radius)  # :1693
    termination_dist = (sphere_pts - pos).norm(dim=-1)  # :1697
    out = ddf_fn(sphere_pts, -dd)  # :1702-1718
    dist = torch.clamp((pos - sphere_pts).norm(dim=-1), max=radius * 2.0)  # :1724-1727
    difference = dist - out["expected_termination_dist"]  # :1730
    visibility = 1.0 - torch.sigmoid(scale * (difference - threshold))  # :1739-1740
    total = torch.ones(R, D, dtype=visibility.dtype) if lower_vis else torch.zeros(R, D, dtype=visibility.dtype)
    if only_upper:
        total = total.masked_scatter(mask[None].expand(R, D), visibility)  # :1745-1753
    else:
        total = visibility.view(R, D)
    res = dict(out)
    res.update(visibility=total, difference=difference, termination_dist=termination_dist, sphere_points=sphere_pts,
               surface_points=pos, upper_mask=mask)
    return res


# =====================================================================================
# A10  proposal sampling [UNPINNED external: nerfstudio]
# =====================================================================================
def sphere_collider(origins: Tensor, directions: Tensor, radius: float = 1.0, near_plane: float = 0.05):
    """nerfstudio SphereCollider (neusky_model.py:213): nears = max(near_plane, t0), fars = t1 of the
    unit sphere; rays that miss get (near_plane, near_plane + 1e-3)-like degenerate spans."""
    a = (directions * directions).sum(-1)
    b = 2 * (origins * directions).sum(-1)
    c = (origins * origins).sum(-1) - radius**2
    disc = b * b - 4 * a * c
    ok = disc > 0
    sq = torch.sqrt(torch.where(ok, disc, torch.zeros_like(disc)))
    t0 = (-b - sq) / (2 * a)
    t1 = (-b + sq) / (2 * a)
    nears = torch.clamp(torch.where(ok, t0, torch.zeros_like(t0)), min=near_plane)
    fars = torch.where(ok, t1, torch.zeros_like(t1))
    fars = torch.maximum(fars, nears + 1e-6)
    return nears[:, None], fars[:, None]


def uniform_bins(nears: Tensor, fars: Tensor, num_samples: int, jitter: Optional[Tensor]):
    """nerfstudio SpacedSampler/UniformSampler (single_jitter): bins in [0,1] -> euclidean.
    jitter [R,1] in [0,1) or None (eval: no perturbation).  Returns spacing bins [R,n+1], euclid bins."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1, dtype=nears.dtype)[None]
    if jitter is not None:
        centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
        upper = torch.cat([centers, bins[..., -1:]], -1)
        lower = torch.cat([bins[..., :1], centers], -1)
        bins = lower + (upper - lower) * jitter
    else:
        bins = bins.expand(nears.shape[0], -1)
    return bins, bins * fars + (1 - bins) * nears


def pdf_sample_bins(existing_bins: Tensor, weights: Tensor, num_samples: int, jitter: Optional[Tensor],
                    histogram_padding: float = 0.01, eps: float = 1e-5):
    """nerfstudio PDFSampler.generate_ray_samples (single_jitter, stratified when jitter given).
    existing_bins [R,n0+1] (spacing domain), weights [R,n0].  Returns new spacing bins [R,num_samples+1]
    and the searchsorted indices `inds` [R,num_samples+1] (int64: the BIT-EXACT ray-sample indices)."""
    num_bins = num_samples + 1
    w = weights + histogram_padding
    wsum = torch.cumsum(w, dim=-1)[..., -1:]  # sequential left-to-right sum (defined order -> reproducible indices)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, dtype=cdf.dtype)
    if jitter is not None:
        u = u[None] + jitter / num_bins
    else:
        u = (u + 1.0 / (2 * num_bins))[None].expand(cdf.shape[0], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, side="right")
    below = torch.clamp(inds - 1, 0, existing_bins.shape[-1] - 1)
    above = torch.clamp(inds, 0, existing_bins.shape[-1] - 1)
    cdf_g0, bins_g0 = torch.gather(cdf, -1, below), torch.gather(existing_bins, -1, below)
    cdf_g1, bins_g1 = torch.gather(cdf, -1, above), torch.gather(existing_bins, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf_g0) / (cdf_g1 - cdf_g0), 0), 0, 1)
    return (bins_g0 + t * (bins_g1 - bins_g0)).detach(), inds


def proposal_density(x: Tensor, p: Dict[str, Tensor], prefix: str, cfg: HashGridCfg,
                     contraction_order: float = float("inf")) -> Tensor:
    """nerfstudio HashMLPDensityField.get_density: contract -> [0,1] -> selector -> hash -> MLP(ReLU) -> trunc_exp."""
    pos = (scene_contraction(x, contraction_order) + 2.0) / 4.0
    sel = ((pos > 0.0) & (pos < 1.0)).all(-1)
    pos = pos * sel[:, None]
    h = hash_grid_encode(pos, p[prefix + "table"], cfg)
    h = F.relu(F.linear(h, p[prefix + "w0"], p[prefix + "b0"]))
    d = F.linear(h, p[prefix + "w1"], p[prefix + "b1"])
    return _TruncExp.apply(d) * sel[:, None]


class _TruncExp(torch.autograd.Function):
    """nerfstudio trunc_exp: forward exp(x); backward g * exp(clamp(x, -15, 15))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(ctx.saved_tensors[0].clamp(-15, 15))


def weights_from_density(density: Tensor, deltas: Tensor) -> Tensor:
    """nerfstudio RaySamples.get_weights: density/deltas [R,S,1]."""
    dd = deltas * density
    alphas = 1 - torch.exp(-dd)
    T = torch.cumsum(dd[..., :-1, :], dim=-2)
    T = torch.exp(-torch.cat([torch.zeros_like(T[..., :1, :]), T], dim=-2))
    return torch.nan_to_num(alphas * T)


def proposal_sample(origins: Tensor, directions: Tensor, nears: Tensor, fars: Tensor, p: Dict[str, Tensor],
                    prop_cfgs: Sequence[HashGridCfg], num_prop: Sequence[int], num_final: int,
                    jitters: Optional[Sequence[Tensor]], anneal: float = 1.0):
    """nerfstudio ProposalNetworkSampler.generate_ray_samples (called neusky_model.py:561).
    jitters: one [R,1] uniform per level (len(num_prop)+1) or None.  Returns dict with final
    spacing/euclid bins, per-level weights/spacing-bins (for interlevel loss) and searchsorted inds."""
    levels = len(num_prop)
    weights_list, sbins_list, inds_list = [], [], []
    sbins = ebins = weights = None
    for lvl in range(levels + 1):
        n = num_prop[lvl] if lvl < levels else num_final
        jit = None if jitters is None else jitters[lvl]
        if lvl == 0:
            sbins, ebins = uniform_bins(nears, fars, n, jit)
        else:
            sbins, inds = pdf_sample_bins(sbins, torch.pow(weights, anneal), n, jit)
            inds_list.append(inds)
            ebins = sbins * fars + (1 - sbins) * nears
        if lvl < levels:
            mid = (ebins[:, :-1] + ebins[:, 1:]) / 2  # frustums.get_positions(): (starts+ends)/2
            x = origins[:, None, :] + directions[:, None, :] * mid[..., None]
            dens = proposal_density(x.reshape(-1, 3), p, f"prop{lvl}.", prop_cfgs[lvl]).view(x.shape[0], n, 1)
            weights = weights_from_density(dens, (ebins[:, 1:] - ebins[:, :-1])[..., None])[..., 0]
            weights_list.append(weights)
            sbins_list.append(sbins)
    return dict(sbins=sbins, ebins=ebins, weights_list=weights_list, sbins_list=sbins_list, inds_list=inds_list)


# =====================================================================================
# A8  illumination (index plumbing PINNED by G9; decoder UNPINNED - ns_reni source absent)
# =====================================================================================
def reni_invariant_inputs(latents: Tensor, dirs: Tensor):
    """THIS PROJECT'S definition of the RENI++ SO(2)-about-z invariant representation (SURVEY App. A.9,
    source absent -> parity unpinned).  latents Z [B,L,3], dirs d [B,3] ->
      cond  [B, 3L]: per latent row (|Z_xy|, Z_z, Z_xy . d_xy)
      x     [B, 2 ]: (|d_xy|, d_z).
    Rotating Z and d together about z leaves both unchanged (property-tested)."""
    zxy, zz = latents[..., :2], latents[..., 2]
    dxy, dz = dirs[..., :2], dirs[..., 2]
    cond = torch.stack([zxy.norm(dim=-1), zz, (zxy * dxy[:, None, :]).sum(-1)], -1).reshape(latents.shape[0], -1)
    x = torch.stack([dxy.norm(dim=-1), dz], -1)
    return cond, x


def reni_decode(latents: Tensor, dirs: Tensor, scale: Tensor, p: Dict[str, Tensor]) -> Tensor:
    """RENI++-shaped FiLM-SIREN decoder (north star: 'RENI++ SIREN illumination decode'):
    x = cat(inv_dir, NeRF2(inv_dir)) (10-d), cond = invariants (3L-d) -> FiLM-SIREN -> 3 log-HDR channels,
    times per-image scale; `unnormalise` = exp (log-domain HDR)."""
    cond, x = reni_invariant_inputs(latents, dirs)
    x = torch.cat([x, nerf_encoding(x, 2, 0.0, 2.0, False)], -1)
    rgb = film_siren(x, cond, p, prefix="reni.")
    return torch.exp(rgb) * scale[:, None]


def reni_attention_decode(latents: Tensor, dirs: Tensor, scale: Tensor, p: Dict[str, Tensor], heads: int = 8, layers: int = 6) -> Tensor:
    """THIS PROJECT'S restatement of the RENI++ attention-conditioned decoder (neusky_config.py:78-95: conditioning="Attention", VN /
    SO2-about-z invariance, 8 heads x 6 layers, hidden 128; ns_reni source absent -> PARITY UNPINNED), in its plain per-pair form:
    latents Z [B,L,3], dirs d [B,3], scale [B] -> HDR radiance [B,3].
      token n:  t_n = E [z_xy . d_xy, z_x d_y - z_y d_x, z_z, |z_xy|] + e   (invariant under a joint rotation of Z and d about z)
      query:    q = X [|d_xy|, d_z, NeRF2(|d_xy|, d_z)] + x
      layer:    q += Wo MHA(LN1(q); K = Wk t + bk, V = Wv t + bv);  q += W2 relu(W1 LN2(q) + b1) + b2;  out = Wout LNf(q) + bout
    The product forms every (pair, token) explicitly here; neusky_amd's AttentionDecoder uses the tokens' linearity in (d_x, d_y)."""
    B, L, _ = latents.shape
    zx, zy, zz = latents[..., 0], latents[..., 1], latents[..., 2]
    dx, dy = dirs[:, 0:1], dirs[:, 1:2]
    inv = torch.stack([zx * dx + zy * dy, zx * dy - zy * dx, zz, torch.sqrt(zx * zx + zy * zy + 1e-20)], -1)  # [B,L,4]
    t = F.linear(inv, p["reni.attn.token_w"], p["reni.attn.token_b"])  # [B,L,H]
    x = torch.stack([torch.sqrt(dirs[:, 0] ** 2 + dirs[:, 1] ** 2 + 1e-20), dirs[:, 2]], -1)
    x = torch.cat([x, nerf_encoding(x, 2, 0.0, 2.0, False)], -1)
    q = F.linear(x, p["reni.attn.query_w"], p["reni.attn.query_b"])  # [B,H]
    H = q.shape[1]
    dh = H // heads
    ln = lambda v, k: F.layer_norm(v, (H,), p[k + "_w"], p[k + "_b"])  # noqa: E731
    for l in range(layers):
        k_ = f"reni.attn.l{l}."
        Q = F.linear(ln(q, k_ + "ln1"), p[k_ + "wq_w"], p[k_ + "wq_b"]).reshape(B, heads, 1, dh)
        K = F.linear(t, p[k_ + "wk_w"], p[k_ + "wk_b"]).reshape(B, L, heads, dh).transpose(1, 2)
        V = F.linear(t, p[k_ + "wv_w"], p[k_ + "wv_b"]).reshape(B, L, heads, dh).transpose(1, 2)
        a = torch.softmax(torch.matmul(Q, K.transpose(-1, -2)) / dh ** 0.5, -1)  # [B,heads,1,L]
        o = torch.matmul(a, V).reshape(B, H)
        q = q + F.linear(o, p[k_ + "wo_w"], p[k_ + "wo_b"])
        q = q + F.linear(torch.relu(F.linear(ln(q, k_ + "ln2"), p[k_ + "ff1_w"], p[k_ + "ff1_b"])), p[k_ + "ff2_w"], p[k_ + "ff2_b"])
    rgb = F.linear(ln(q, "reni.attn.lnf"), p["reni.attn.out_w"], p["reni.attn.out_b"])
    return torch.exp(rgb) * scale[:, None]


def sample_illumination(cam_idx: Tensor, ray_dirs: Tensor, dirs: Tensor, latents: Tensor, scales: Tensor, decode_fn):
    """neusky/models/neusky_model.py:445-551 on compact inputs.  cam_idx [R] (every sample of a ray has the
    ray's camera index), ray_dirs [R,3], dirs [D,3].  decode_fn(latents[B,L,3], dirs[B,3], scale[B]) -> [B,3].
    Returns (cam_colours [U,D,3], inverse [R] so that colours_of_ray = cam_colours[inverse], bg [R,3]).
    The reference's [R*S,D,3] tensors are cam_colours[inverse] repeated over S (:512-518)."""
    unique, inverse = torch.unique(cam_idx, return_inverse=True)  # :461-463 (sorted unique)
    U, D = unique.shape[0], dirs.shape[0]
    ci = unique[:, None].expand(U, D).reshape(-1)  # :466-476
    dd = dirs[None].expand(U, D, 3).reshape(-1, 3)  # :470-479
    cols = decode_fn(latents[ci], dd, scales[ci]).reshape(U, D, 3)  # :488-510
    bg = decode_fn(latents[cam_idx], ray_dirs, scales[cam_idx])  # :535-549
    return cols, inverse, bg


# =====================================================================================
# A12  losses (PINNED by G7 / G6 where in-tree)
# =====================================================================================
def sky_pixel_loss(inputs: Tensor, targets: Tensor, mask: Tensor, alpha: float) -> Tensor:
    """neusky/model_components/losses.py:44-58."""
    inputs, targets = inputs * mask, targets * mask
    mse = F.mse_loss(inputs, targets)
    sim = F.cosine_similarity(inputs, targets, dim=1, eps=1e-20)
    return mse + alpha * (1 - sim.mean())


def monosdf_normal_loss(normal_pred: Tensor, normal_gt: Tensor) -> Tensor:
    """nerfstudio monosdf_normal_loss [UNPINNED] (called neusky_model.py:1000)."""
    normal_gt = F.normalize(normal_gt, p=2, dim=-1)
    normal_pred = F.normalize(normal_pred, p=2, dim=-1)
    return torch.abs(normal_pred - normal_gt).sum(dim=-1).mean() + (1.0 - (normal_pred * normal_gt).sum(-1)).mean()


def _outer(t0_starts, t0_ends, t1_starts, t1_ends, y1):
    cy1 = torch.cat([torch.zeros_like(y1[..., :1]), torch.cumsum(y1, dim=-1)], dim=-1)
    idx_lo = torch.clamp(torch.searchsorted(t1_starts.contiguous(), t0_starts.contiguous(), side="right") - 1, 0, y1.shape[-1] - 1)
    idx_hi = torch.clamp(torch.searchsorted(t1_ends.contiguous(), t0_ends.contiguous(), side="right"), 0, y1.shape[-1] - 1)
    return torch.take_along_dim(cy1[..., 1:], idx_hi, dim=-1) - torch.take_along_dim(cy1[..., :-1], idx_lo, dim=-1)


def interlevel_loss(weights_list: Sequence[Tensor], sbins_list: Sequence[Tensor]) -> Tensor:
    """nerfstudio interlevel_loss / lossfun_outer [UNPINNED] (called neusky_model.py:987-988).
    weights_list[i] [R,n_i], sbins_list[i] [R,n_i+1]; last entry = final NeuS samples."""
    c, w = sbins_list[-1].detach(), weights_list[-1].detach()
    loss = 0.0
    for sb, wp in zip(sbins_list[:-1], weights_list[:-1]):
        w_outer = _outer(c[..., :-1], c[..., 1:], sb[..., :-1], sb[..., 1:], wp)
        loss = loss + torch.mean(torch.clip(w - w_outer, min=0) ** 2 / (w + 1e-7))
    return loss


NEUSKY_LOSS_COEFFICIENTS = {  # neusky/configs/neusky_config.py:127-141
    "rgb_l1_loss": 1.0, "rgb_l2_loss": 0.0, "cosine_colour_loss": 1.0, "eikonal loss": 0.1, "fg_mask_loss": 1.0,
    "normal_loss": 1.0, "depth_loss": 1.0, "sdf_level_set_visibility_loss": 1.0, "interlevel_loss": 1.0,
    "sky_pixel_loss": 1.0, "hashgrid_density_loss": 1e-4, "ground_plane_loss": 0.1, "visibility_sigmoid_loss": 0.01,
}
DDF_LOSS_COEFFICIENTS = {  # neusky/configs/neusky_config.py:188-197
    "depth_l1_loss": 1.0, "depth_l2_loss": 0.0, "sdf_l1_loss": 1.0, "sdf_l2_loss": 0.01, "prob_hit_loss": 0.01,
    "normal_loss": 1.0, "multi_view_loss": 0.01, "sky_ray_loss": 1.0,
}


def scale_dict(d: Dict[str, Tensor], coefficients: Dict[str, float]) -> Dict[str, Tensor]:
    """nerfstudio misc.scale_dict [UNPINNED]: only keys present in `coefficients` are scaled.  Note the
    reference stores the eikonal term under 'eikonal_loss' (neusky_model.py:960) while its coefficient is
    keyed 'eikonal loss' (neusky_config.py:131): the 0.1 is therefore never applied - reproduced."""
    return {k: (v * coefficients[k] if k in coefficients else v) for k, v in d.items()}


def neusky_losses(out: Dict[str, Tensor], image: Tensor, mask: Tensor, threshold: Tensor,
                  target_min_bias: float = 0.1, sky_alpha: float = 0.1) -> Dict[str, Tensor]:
    """neusky/models/neusky_model.py:933-1035 (train branch, `neusky` config inclusions :102-126),
    UNSCALED (apply scale_dict).  out: rgb [R,3], eik_grad [R,S,3], weights [R,S,1], normal [R,3],
    hdr_background_colours [R,3], grid_density [G,1], sdf_at_termination [M,1] (+ weights_list/sbins_list)."""
    fg, ground, sky = mask[..., 1], mask[..., 2], mask[..., 3]
    ld: Dict[str, Tensor] = {}
    keep = (1 - sky.to(image.dtype))[:, None]
    ld["rgb_l1_loss"] = F.l1_loss(image * keep, out["rgb"] * keep)  # :947-950
    ld["eikonal_loss"] = ((out["eik_grad"].norm(2, dim=-1) - 1) ** 2).mean()  # :958-960
    ws = torch.nan_to_num(out["weights"].sum(dim=1).clip(1e-3, 1.0 - 1e-3), nan=0.5)  # :964-965
    ld["fg_mask_loss"] = F.binary_cross_entropy(ws, fg.to(ws.dtype)[:, None])  # :966-967
    if "weights_list" in out:
        ld["interlevel_loss"] = interlevel_loss(out["weights_list"], out["sbins_list"])  # :987-988
    ld["hashgrid_density_loss"] = out["grid_density"].abs().mean()  # :990-993
    gm = ground.to(image.dtype)[:, None].expand_as(out["normal"])
    ngt = torch.tensor([0.0, 0.0, 1.0], dtype=image.dtype).expand_as(out["normal"])
    ld["ground_plane_loss"] = monosdf_normal_loss(out["normal"] * gm, ngt * gm)  # :995-1000
    sm = sky.to(image.dtype)[:, None].expand(-1, 3)
    ld["sky_pixel_loss"] = sky_pixel_loss(linear_to_srgb(out["hdr_background_colours"]), image, sm, sky_alpha)  # :1002-1009
    ld["visibility_sigmoid_loss"] = (threshold - target_min_bias) ** 2  # :1011-1030 (bias only, scale fixed)
    ld["sdf_level_set_visibility_loss"] = (out["sdf_at_termination"] ** 2).mean()  # :1032-1035
    return ld


def neusky_eval_losses(rgb: Tensor, hdr_bg: Tensor, image: Tensor, mask: Tensor, sky_alpha: float = 0.1, rgb_l2: bool = False,
                       cosine_colour: bool = False, sky_pixel: bool = True) -> Dict[str, Tensor]:
    """neusky/models/neusky_model.py:1036-1059: the evaluation / eval-latent-fitting branch of get_loss_dict, UNSCALED (PINNED by
    G7's `evalloss_*` / `evalall_*` / `evalosr_*` vectors).  The `neusky` config includes rgb_l1 and the sky-pixel term only
    (neusky_config.py:102-126); sky_pixel=False is the 'nerf_osr_envmap' method (:1048)."""
    sky = mask[..., 3].to(image.dtype)
    keep = (1 - sky)[:, None]
    im, pred = image * keep, rgb * keep
    ld: Dict[str, Tensor] = {"rgb_l1_loss": F.l1_loss(im, pred)}  # :1038-1043
    if rgb_l2:
        ld["rgb_l2_loss"] = F.mse_loss(im, pred)  # :1044-1045
    if cosine_colour:
        ld["cosine_colour_loss"] = torch.mean(1 - F.cosine_similarity(im, pred, dim=1))  # :1046-1048
    if sky_pixel:
        ld["sky_pixel_loss"] = sky_pixel_loss(linear_to_srgb(hdr_bg), image, sky[:, None].expand(-1, 3), sky_alpha)  # :1051-1058
    return ld


def ddf_losses(expected: Tensor, gt_term: Tensor, mask: Tensor, distance_weight: Tensor, sdf_at_term: Tensor,
               mv_expected: Tensor, mv_gt: Tensor, sky_expected: Tensor, sky_gt: Tensor) -> Dict[str, Tensor]:
    """neusky/models/ddf_model.py:407-493 with the `neusky` config (:178-205): mask_to_circumference=False,
    centre-weighted depth L1 (:427-433), sdf L2 (:457-461), multi-view hinge^2 (:475-483), sky-ray L1 (:485-490).
    expected [M], gt_term/mask [M,1], distance_weight [M], sdf_at_term [M,1]; UNSCALED."""
    e = expected[:, None] * mask
    g = gt_term * mask
    ld = {"depth_l1_loss": torch.mean(torch.abs(e - g) * distance_weight[:, None])}
    ld["sdf_l2_loss"] = F.mse_loss(sdf_at_term * mask, torch.zeros_like(sdf_at_term))
    ld["multi_view_loss"] = torch.mean(F.relu(mv_expected - mv_gt) ** 2)  # sic: [M] - [M,1] broadcasts to [M,M]
    ld["sky_ray_loss"] = F.l1_loss(sky_expected, sky_gt)
    return ld


def ddf_model_outputs(positions: Tensor, directions: Tensor, gt_term: Tensor, mv_points: Tensor, sky_o: Tensor,
                      sky_d: Tensor, radius: float, ddf_fn, sdf_fn, exp: float = 3.0):
    """neusky/models/ddf_model.py:183-369 (training, `neusky` config).  ddf_fn(sphere_pos, world_dir) -> [M];
    sdf_fn(x [M,3]) -> [M,1]; mv_points are the explicit 'random other positions' (:289-294, z made >= 0)."""
    out = {"expected_termination_dist": ddf_fn(positions, directions)}  # :217-219
    dist = positions[..., :2].norm(dim=-1) / radius  # :232-235
    out["distance_weight"] = 1.0 - dist**exp  # :237
    out["sdf_at_termination"] = sdf_fn(positions + directions * out["expected_termination_dist"][:, None])  # :243-251
    gt_pts = positions + directions * gt_term  # :286
    pts = mv_points.clone()
    pts[:, 2] = pts[:, 2].abs()  # :294
    dvec = gt_pts - pts
    dlen = dvec.norm(dim=-1)
    out["multi_view_termintation_dist"] = gt_term  # :321 (sic)
    out["multi_view_expected_termination_dist"] = ddf_fn(pts, dvec / dlen[:, None])  # :297-322
    sp = ray_sphere_intersection_free(sky_o, sky_d, radius)  # :337-339
    out["sky_ray_termination_dist"] = (sky_o - sp).norm(dim=-1)  # :343
    out["sky_ray_expected_termination_dist"] = ddf_fn(sp, -sky_d)  # :346-363
    return out


# =====================================================================================
# A13  vMF sampler (statistical parity only, G10)
# =====================================================================================
def vmf_cos(kappa: float, n: int, gen: torch.Generator, d: int = 3) -> Tensor:
    """neusky/model_components/ddf_sampler.py:205-223 (Wood's rejection sampler)."""
    b = (d - 1) / (2 * kappa + (4 * kappa**2 + (d - 1) ** 2) ** 0.5)
    x0 = (1 - b) / (1 + b)
    c = kappa * x0 + (d - 1) * math.log(1 - x0**2)
    out, found = [], 0
    beta = torch.distributions.beta.Beta((d - 1) / 2, (d - 1) / 2)
    while found < n:
        m = min(n, int((n - found) * 1.5))
        # Beta((d-1)/2,(d-1)/2) with d=3 is U(0,1); sample through the generator for determinism
        z = torch.rand(m, generator=gen, dtype=torch.float64) if d == 3 else beta.sample((m,)).double()
        t = (1 - (1 + b) * z) / (1 - (1 - b) * z)
        test = kappa * t + (d - 1) * torch.log(1 - x0 * t) - c
        acc = test >= -math.e  # sic: -exp(1) (:220)
        out.append(t[acc]); found += int(acc.sum())
    return torch.cat(out)[:n]


def vmf_ddf_rays(num_positions: int, num_directions: int, kappa: float, radius: float, gen: torch.Generator):
    """neusky/model_components/ddf_sampler.py:225-286: upper-hemisphere sphere points, vMF(kappa) inward dirs."""
    theta = 2 * math.pi * torch.rand(num_positions, generator=gen, dtype=torch.float64)
    phi = torch.acos(2 * torch.rand(num_positions, generator=gen, dtype=torch.float64) - 1)
    pos = torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], 1)
    pos = torch.where(pos[:, 2:3] < 0, -pos, pos)  # :255-257
    normals = -pos
    z = torch.randn(num_positions, num_directions, 3, generator=gen, dtype=torch.float64)
    z = z / z.norm(dim=-1, keepdim=True)
    z = z - torch.einsum("nij,nj->ni", z, normals)[..., None] * normals[:, None, :]
    z = z / z.norm(dim=-1, keepdim=True)
    cos = vmf_cos(kappa, num_positions * num_directions, gen).reshape(num_positions, num_directions)
    sin = torch.sqrt(1 - cos**2)
    x = z * sin[..., None] + cos[..., None] * normals[:, None, :]
    x = x / x.norm(dim=-1, keepdim=True)
    flip = torch.einsum("nij,nj->ni", x, normals) < 0  # :262-266
    x = torch.where(flip[..., None], -x, x)
    P = (pos * radius)[:, None, :].expand(-1, num_directions, -1).reshape(-1, 3)
    return P, x.reshape(-1, 3)


# =====================================================================================
# full train-step forward (composition of the rows above; neusky_pipeline.py:241-291)
# =====================================================================================
@dataclass
class StepCfg:
    num_prop: Tuple[int, ...] = (256, 96)
    num_final: int = 48
    radius: float = 1.0
    sigmoid_scale: float = 25.0
    anneal: float = 1.0
    grid_res: int = 10
    field_grid: HashGridCfg = field(default_factory=lambda: HashGridCfg(smoothstep=True))
    ddf_grid: HashGridCfg = field(default_factory=lambda: HashGridCfg(smoothstep=False))
    prop_grids: Tuple[HashGridCfg, ...] = field(default_factory=lambda: (
        HashGridCfg(n_levels=5, log2_hashmap_size=17, max_res=64), HashGridCfg(n_levels=5, log2_hashmap_size=17, max_res=256)))


def render_depth(weights: Tensor, ebins: Tensor) -> Tensor:
    """nerfstudio DepthRenderer('expected') as called at neusky_model.py:591: weights [R,S] -> [R,1]"""
    mid = (ebins[:, :-1] + ebins[:, 1:]) / 2
    depth = (weights * mid).sum(-1, keepdim=True) / (weights.sum(-1, keepdim=True) + 1e-10)
    return torch.clip(depth, mid.min(), mid.max())


def field_pass(p, cfg: StepCfg, origins, directions, ebins):
    R, S = ebins.shape[0], ebins.shape[1] - 1
    o = origins[:, None, :].expand(R, S, 3)
    d = directions[:, None, :].expand(R, S, 3)
    fo = sdf_field_outputs(o, d, ebins[:, :-1, None], ebins[:, 1:, None], p, cfg.field_grid)
    w, T = weights_from_alphas(fo["alpha"])
    fo["weights"], fo["bg_transmittance"] = w[..., 0], T[:, -1]
    return fo


def neusky_forward_rays(p: Dict[str, Tensor], cfg: StepCfg, origins: Tensor, directions: Tensor, cam_idx: Tensor, jitters,
                        light_dirs: Tensor):
    """The per-ray part of the training forward (neusky_model.py:553-631, 797-805): proposal sampling, field, illumination,
    depth, DDF visibility (+ the sdf probe at the predicted termination points) and the Lambertian render.  Every output row
    depends on its own ray only (its camera's latents, its jitters, the shared direction set), so a SLICE of a large batch can be
    checked on its own (tests/test_gpu_full_size.py).  -> (samp, field outputs, bg, p2p, visibility dict, rgb)"""
    nears, fars = sphere_collider(origins, directions, cfg.radius)  # neusky_model.py:440-441
    samp = proposal_sample(origins, directions, nears, fars, p, cfg.prop_grids, cfg.num_prop, cfg.num_final, jitters, cfg.anneal)
    ebins = samp["ebins"]
    fo = field_pass(p, cfg, origins, directions, ebins)  # :563-568
    weights = fo["weights"]
    decode = lambda lat, dd, sc: reni_decode(lat, dd, sc, p)
    cols, inverse, bg = sample_illumination(cam_idx, directions, light_dirs, p["train_latents"], p["train_scale"], decode)  # :573
    p2p = render_depth(weights, ebins)  # :591
    ddf_fn = lambda sp, dd: ddf_query(sp, dd, p, cfg.ddf_grid, cfg.radius)

    def vis_field(sp, dd):
        t = ddf_fn(sp, dd)
        return {"expected_termination_dist": t, "sdf_at_termination": sdf_at_positions(sp + dd * t[:, None], p, cfg.field_grid)}

    vis = compute_visibility(origins, directions, p2p.detach(), light_dirs, p["visibility_threshold"], cfg.sigmoid_scale,
                             cfg.radius, vis_field, True, True)  # :624-630 ('depth' stop-gradient mode)
    rgb = lambertian_render(fo["albedo"], fo["normals"], light_dirs, cols, inverse, vis["visibility"], bg, weights)  # :797-805
    return samp, fo, bg, p2p, vis, rgb


def neusky_train_step(p: Dict[str, Tensor], cfg: StepCfg, origins: Tensor, directions: Tensor, cam_idx: Tensor,
                      image: Tensor, mask: Tensor, rnd: Dict[str, Tensor], light_dirs: Tensor):
    """Forward of one full training step -> (scaled loss dict, outputs).  All randomness is explicit:
    rnd = {jitters: [L+1 x [R,1]], grid_perturb [G,3], grid_dirs [G,3], ddf_rays (o,d), ddf_jitters, mv_points [Mv,3],
    sky_o, sky_d}; light_dirs [D,3] is the (already rotated) illumination direction set."""
    dt = origins.dtype
    R = origins.shape[0]
    samp, fo, bg, p2p, vis, rgb = neusky_forward_rays(p, cfg, origins, directions, cam_idx, rnd["jitters"], light_dirs)
    ebins, weights = samp["ebins"], fo["weights"]
    ddf_fn = lambda sp, dd: ddf_query(sp, dd, p, cfg.ddf_grid, cfg.radius)
    # hash-grid density probe (:672-734): jittered res^3 lattice over the scene box, alpha per axis gap
    res = cfg.grid_res
    lin = torch.linspace(-cfg.radius, cfg.radius, res, dtype=dt)
    X, Y, Z = torch.meshgrid(lin, lin, lin, indexing="ij")
    gap = torch.full((3,), 2 * cfg.radius / res, dtype=dt)
    gpos = torch.stack((X, Y, Z), -1).reshape(-1, 3) + (rnd["grid_perturb"].to(dt) * gap - gap / 2)
    gdir = rnd["grid_dirs"].to(dt)
    gdir = gdir / gdir.norm(dim=-1, keepdim=True)
    xg = gpos.detach().requires_grad_(True)
    hg = geo_network(xg, p, cfg.field_grid)
    gg = torch.autograd.grad(hg[:, :1], xg, torch.ones_like(hg[:, :1]), create_graph=True)[0]
    grid_density = neus_alpha(hg[:, :1], gg, gdir, gap[None, :], p["field.variance"])  # [G,3] (sic: deltas = gap [3])
    out = {
        "rgb": rgb, "eik_grad": fo["gradients"], "weights": weights[..., None], "normal": (weights[..., None] * fo["normals"]).sum(-2),
        "hdr_background_colours": bg, "grid_density": grid_density, "sdf_at_termination": vis["sdf_at_termination"],
        "weights_list": samp["weights_list"] + [weights], "sbins_list": samp["sbins_list"] + [samp["sbins"]],
        "p2p_dist": p2p, "visibility": vis["visibility"], "expected_termination_dist": vis["expected_termination_dist"],
        "pdf_inds_list": samp["inds_list"], "albedo": fo["albedo"], "sdf": fo["sdf"], "ebins": ebins,
    }
    ld = scale_dict(neusky_losses(out, image, mask, p["visibility_threshold"].reshape(())), NEUSKY_LOSS_COEFFICIENTS)
    # DDF fitting (:271-289): second sampler + field pass on the vMF rays, then the DDF model
    do, dd = rnd["ddf_rays"]
    dn, df = sphere_collider(do, dd, cfg.radius)
    dsamp = proposal_sample(do, dd, dn, df, p, cfg.prop_grids, cfg.num_prop, cfg.num_final, rnd["ddf_jitters"], cfg.anneal)
    dfo = field_pass(p, cfg, do, dd, dsamp["ebins"])
    acc = dfo["weights"].sum(-1, keepdim=True)
    dmask = (acc > 0.0).to(dt)  # visibility_accumulation_mask_threshold = 0.0 (neusky_config.py:214)
    gt_term = torch.clamp(render_depth(dfo["weights"], dsamp["ebins"]), max=2 * cfg.radius)  # :1348-1351
    dout = ddf_model_outputs(do, dd, gt_term, rnd["mv_points"].to(dt), rnd["sky_o"], rnd["sky_d"], cfg.radius, ddf_fn,
                             lambda x: sdf_at_positions(x, p, cfg.field_grid))
    dl = ddf_losses(dout["expected_termination_dist"], gt_term, dmask, dout["distance_weight"], dout["sdf_at_termination"],
                    dout["multi_view_expected_termination_dist"], dout["multi_view_termintation_dist"],
                    dout["sky_ray_expected_termination_dist"], dout["sky_ray_termination_dist"])
    ld.update(scale_dict(dl, DDF_LOSS_COEFFICIENTS))
    out.update({"ddf_" + k: v for k, v in dout.items()})
    out["ddf_gt_termination_dist"] = gt_term
    return ld, out


def neusky_render(p: Dict[str, Tensor], cfg: StepCfg, origins: Tensor, directions: Tensor, latent: Tensor, scale: Tensor,
                  light_dirs: Tensor, rotation: Optional[Tensor] = None):
    """Eval-mode forward of a camera ray bundle with ONE illumination latent (relighting render,
    neusky_model.py:1369-1501 -> forward -> get_outputs eval branch :814-879): deterministic samplers (no jitter,
    fixed directions :451-454), optional z-rotation of the illumination (:483-493, render_animation.py:196-207),
    visibility without the sdf probe, Lambertian render with the eval clamp."""
    R = origins.shape[0]
    nears, fars = sphere_collider(origins, directions, cfg.radius)
    samp = proposal_sample(origins, directions, nears, fars, p, cfg.prop_grids, cfg.num_prop, cfg.num_final, None, 1.0)
    fo = field_pass(p, cfg, origins, directions, samp["ebins"])
    weights = fo["weights"]
    ldir = light_dirs if rotation is None else light_dirs @ rotation.T
    rdir = directions if rotation is None else directions @ rotation.T
    D = light_dirs.shape[0]
    cols = reni_decode(latent[None].expand(D, -1, -1), ldir, scale.expand(D), p)[None]  # [1,D,3]
    bg = reni_decode(latent[None].expand(R, -1, -1), rdir, scale.expand(R), p)
    p2p = render_depth(weights, samp["ebins"])
    ddf_fn = lambda sp, dd: {"expected_termination_dist": ddf_query(sp, dd, p, cfg.ddf_grid, cfg.radius)}
    vis = compute_visibility(origins, directions, p2p, light_dirs, p["visibility_threshold"], cfg.sigmoid_scale, cfg.radius,
                             ddf_fn, True, True)
    rgb = lambertian_render(fo["albedo"], fo["normals"], light_dirs, cols, torch.zeros(R, dtype=torch.long), vis["visibility"], bg,
                            weights, training=False)
    return {"rgb": rgb, "p2p_dist": p2p, "accumulation": weights.sum(-1, keepdim=True),
            "normal": (weights[..., None] * fo["normals"]).sum(-2), "albedo": (weights[..., None] * fo["albedo"]).sum(-2) + (1 - weights.sum(-1, keepdim=True)),
            "visibility": vis["visibility"]}


# =====================================================================================
# (f)2  eval-latent fitting: neusky/models/neusky_model.py:1503-1588 (per_image), loss = eval branch of get_loss_dict
#       (:1036-1059): sky-masked rgb L1 + RENISkyPixelLoss, both with coefficient 1 (neusky_config.py:127-141)
# =====================================================================================
def neusky_eval_fit_loss(p: Dict[str, Tensor], cfg: StepCfg, origins: Tensor, directions: Tensor, cam_idx: Tensor, image: Tensor,
                         mask: Tensor, rnd: Dict[str, Tensor], light_dirs: Tensor, latents: Tensor, scales: Tensor) -> Tensor:
    """forward of one fitting step (the model is in training mode: jittered samplers, rotated light directions) with the
    EVAL latents / scales as the only variables; everything that does not depend on them is evaluated without a graph"""
    with torch.no_grad():
        nears, fars = sphere_collider(origins, directions, cfg.radius)
        samp = proposal_sample(origins, directions, nears, fars, p, cfg.prop_grids, cfg.num_prop, cfg.num_final, rnd["jitters"], cfg.anneal)
    ebins = samp["ebins"]
    pc = {k: v.detach() for k, v in p.items()}
    fo = field_pass(pc, cfg, origins, directions, ebins)
    fo = {k: v.detach() for k, v in fo.items()}
    weights = fo["weights"]
    decode = lambda lat, dd, sc: reni_decode(lat, dd, sc, pc)
    cols, inverse, bg = sample_illumination(cam_idx, directions, light_dirs, latents, scales, decode)
    p2p = render_depth(weights, ebins)
    with torch.no_grad():
        vis = compute_visibility(origins, directions, p2p, light_dirs, pc["visibility_threshold"], cfg.sigmoid_scale, cfg.radius,
                                 lambda sp, dd: {"expected_termination_dist": ddf_query(sp, dd, pc, cfg.ddf_grid, cfg.radius)}, True, True)
    rgb = lambertian_render(fo["albedo"], fo["normals"], light_dirs, cols, inverse, vis["visibility"], bg, weights)
    ld = scale_dict(neusky_eval_losses(rgb, bg, image, mask), NEUSKY_LOSS_COEFFICIENTS)  # :1036-1059, both coefficients 1
    return ld["rgb_l1_loss"] + ld["sky_pixel_loss"]


def adam_fit(params, loss_fn, steps: int, lr: float, lr_final: float, eps: float = 1e-15, betas=(0.9, 0.999)):
    """torch.optim.Adam + nerfstudio ExponentialDecayScheduler (no warm-up): lr_t = exp(log lr (1 - t) + log lr_final t), t = it / steps.
    loss_fn(it) -> scalar.  Updates `params` (leaf tensors requiring grad) in place; returns the loss trace."""
    import math
    m = [torch.zeros_like(q) for q in params]
    v = [torch.zeros_like(q) for q in params]
    trace = []
    for it in range(steps):
        loss = loss_fn(it)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        t = min(max(it / steps, 0.0), 1.0)
        lr_t = math.exp(math.log(lr) * (1 - t) + math.log(lr_final) * t)
        with torch.no_grad():
            for q, g, mm, vv in zip(params, grads, m, v):
                if g is None:
                    g = torch.zeros_like(q)
                mm.mul_(betas[0]).add_(g, alpha=1 - betas[0])
                vv.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
                mhat, vhat = mm / (1 - betas[0] ** (it + 1)), vv / (1 - betas[1] ** (it + 1))
                q.sub_(lr_t * mhat / (vhat.sqrt() + eps))
        trace.append(float(loss))
    return trace
