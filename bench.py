"""bench.py - train-step rays/sec of the NeuSky hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full training iteration on a synthetic NeRF-OSR-lk2-shaped batch (BASELINE.json config[2]:
1024 rays/GPU x 96 samples, 512 illumination directions, RENI-shaped illumination decode, DDF visibility):
proposal sampling -> hash encode -> SDF/albedo MLPs -> illumination decode -> DDF hemisphere visibility ->
hemisphere integral + composite -> all losses -> backward -> gradient all-reduce (N>1) -> 5 Adam groups.
Inputs are generated on the device before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch
import torch.distributed as dist

# fp32 matrix-core peak and HBM peak from /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
RAYS, SAMPLES, DIRECTIONS, PROPOSAL = 1024, 96, 512, (256, 96)
STEP_ALGORITHMIC_TFLOP = 2.3  # BASELINE.md section 4 / SURVEY 8(d): forward 762 GFLOP, forward + backward ~ 3 x
PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"  # rocprofv3 --pmc passes of this round's kernels (tools/pmc_bench.sh + tools/pmc_step_traffic.py)


def build_pipeline(device, world_size, local_rank, rays=RAYS, samples=SAMPLES, directions=DIRECTIONS, proposal=PROPOSAL):
    from neusky_amd.configs.neusky_config import synthetic_pipeline_config
    cfg = synthetic_pipeline_config()  # the `neusky` method's pipeline, synthetic lk2-shaped data (no dataset in the image)
    cfg.model.num_neus_samples_per_ray = samples  # BASELINE.json fixes 96
    cfg.model.num_proposal_samples_per_ray = tuple(proposal)
    cfg.model.illumination_sampler.num_directions = directions
    cfg.datamanager.train_num_rays_per_batch = rays
    pipe = cfg.setup(device=device, world_size=world_size, local_rank=local_rank)
    pipe.train()
    return pipe


class KernelTimer:
    """HIP-event timing of every launch of the step's heavy kernel families on torch's current stream (the stream the C-ABI
    launches on), during ONE eager iteration: the fused FiLM-SIREN chain kernels (forward / FiLM backward / mapping backward,
    per hidden width) and the dense-layer kernels (LDS-DMA planes kernel, register-staged split kernel, by precision).
    Each record carries the ALGORITHMIC FLOPs of the launch (2 M N K of the contraction the layer needs; for the chain
    kernels the products of the layers they replace) and, for the FiLM backward, the FLOPs it actually executes (it re-forms
    the frequency / phase tiles instead of reading a [M, 2 n H] matrix back)."""

    def __init__(self):
        self.records = {}

    def _timed(self, key, fn, flops, executed, *a, nbytes=None, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **kw)
        e1.record()
        self.records.setdefault(key, []).append((e0, e1, flops, executed, nbytes))
        return out

    def install(self):
        from neusky_amd import hip
        self._orig = {n: getattr(hip, n) for n in ("gemm", "gemm_planes", "film_chain_fwd", "film_chain_bwd_film", "film_chain_bwd_map",
                                                   "wgrad_native_batch", "field_geo_fwd", "field_colour_fwd", "field_colour_bwd", "field_geo_bwd",
                                                   "encode_fwd", "encode_bwd", "sdf_chain_fwd", "sdf_chain_bwd", "native_weighted_colsum",
                                                   "adam_step")}
        o, t = self._orig, self

        def gemm(A, B, Cout, M, N, K, **kw):
            prec = int(kw.get("precision", 0))
            kind = "weight-gradient" if not kw.get("a_kcontig", True) else "layer"
            key = f"gemm_bf16s_kernel ({kind}, precision {prec})" if (prec and N > 64) else f"gemm_f32_kernel ({kind})"
            return t._timed(key, o["gemm"], 2.0 * M * N * K, 2.0 * M * N * K, A, B, Cout, M, N, K, **kw)

        def gemm_planes(A, planes, Cout, M, N, K, **kw):
            key = f"gemm_planes_kernel (precision {int(kw.get('precision', 0))})"
            return t._timed(key, o["gemm_planes"], 2.0 * M * N * K, 2.0 * M * N * K, A, planes, Cout, M, N, K, **kw)

        def dims(net):
            return net.hidden, net.n_map, net.n_film, net.cond_dim, net.x_dim, net.out_dim

        def fwd(net, stream, table, cond, x, M, *a, **kw):
            H, nm, nf, cd, xd, od = dims(net)
            fl = 2.0 * M * (cd * H + (nm - 1) * H * H + H * 2 * nf * H + xd * H + (nf - 1) * H * H + H * od)
            # algorithmic bytes: the input rows in, the saved activations (when kept) and the head output out, each once
            saves = (nm + 2 * nf) if (len(a) > 0 and a[0] is not None) else 0
            by = 4.0 * M * (cond.shape[1] + x.shape[1] + saves * H + 4)
            return t._timed(f"film_fwd_kernel<{H}>", o["film_chain_fwd"], fl, fl, net, stream, table, cond, x, M, *a, nbytes=by, **kw)

        def bwd_film(net, stream, table, M, *a, **kw):
            H, nm, nf, cd, xd, od = dims(net)
            fl = 2.0 * M * ((nf - 1) * H * H + H * xd)
            # algorithmic bytes: d_res, h_last and the z saves in; dz and dF / dphase out
            by = 4.0 * M * (4 + H + nf * H + nf * H + 2 * nf * H + 16)
            return t._timed(f"film_bwd4_kernel<{H}>", o["film_chain_bwd_film"], fl, fl + 2.0 * M * 2 * nf * H * H, net, stream, table, M, *a,
                            nbytes=by, **kw)

        def bwd_map(net, stream, table, M, *a, **kw):
            H, nm, nf, cd, xd, od = dims(net)
            fl = 2.0 * M * (2 * nf * H * H + (nm - 1) * H * H + H * cd)
            by = 4.0 * M * (2 * nf * H + nm * H + nm * H + cd)  # dF / dphase and the h saves in, dpre and d_cond out
            return t._timed(f"film_bwd_map_kernel<{H}>", o["film_chain_bwd_map"], fl, fl, net, stream, table, M, *a, nbytes=by, **kw)

        def wgrad(problems, rows):
            wa = lambda q: q.width_a if (q.lda > 0 or q.width_a > 0) else 32 * q.nnt_a  # noqa: E731
            wb = lambda q: q.width_b if (q.ldb > 0 or q.width_b > 0) else 32 * q.nnt_b  # noqa: E731
            fl = sum(2.0 * rows * wa(q) * wb(q) for q in problems)
            # algorithmic bytes: every operand matrix once (the layer input shared by the blocks of a wide layer counts once)
            seen, by = set(), 0.0
            for q in problems:
                for p_, w in ((q.dZ, wa(q)), (q.X, wb(q))):
                    if p_ not in seen:
                        seen.add(p_)
                        by += 4.0 * rows * w
            # ONE family for every launch of the kernel (DDF chain, illumination chain, field layers, sdf probe), as rocprof counts it
            return t._timed("wgrad_native_kernel", o["wgrad_native_batch"], fl, fl, problems, rows, nbytes=by)

        # fused SDF / albedo field (csrc/field_chain.hip): products of the layers they replace; bytes = inputs, saves and outputs once each
        def geo_fwd(net, pack, ET, N, a0q, a1q, Eq, *a, **kw):
            fl = 2.0 * 4 * N * (net.in_dim * 256 + 256 * 256 + 256)
            by = 4.0 * 4 * N * (ET.shape[1] + 256 + 256 + (128 if Eq is not None else 0)) + 4.0 * N * 5
            return t._timed("field_geo_fwd_kernel", o["field_geo_fwd"], fl, fl, net, pack, ET, N, a0q, a1q, Eq, *a, nbytes=by, **kw)

        def col_fwd(net, pack, ET, N, *a, **kw):
            fl = 2.0 * N * (256 * 256 + 300 * 256 + 256 * 256 + 3 * 256)
            by = 4.0 * N * (256 + 40 + 256 * 4 + 128 + 4)
            return t._timed("field_colour_fwd_kernel", o["field_colour_fwd"], fl, fl, net, pack, ET, N, *a, nbytes=by, **kw)

        def col_bwd(net, pack, N, *a, **kw):
            fl = 2.0 * N * (256 * 256 + 300 * 256 + 256 * 256 + 3 * 256)
            by = 4.0 * N * (3 + 4 + 256 * 2 + 4 + 256 * 4 + 40)
            return t._timed("field_colour_bwd_kernel", o["field_colour_bwd"], fl, fl, net, pack, N, *a, nbytes=by, **kw)

        def geo_bwd(net, pack, N, g_sdf, g_grad, da1v, *a, **kw):
            fl = 2.0 * 4 * N * (256 * 256 + net.in_dim * 256)
            by = 4.0 * 4 * N * (256 * 4 + 72) + 4.0 * N * (4 + (256 + 40 if da1v is not None else 0))
            return t._timed("field_geo_bwd_kernel", o["field_geo_bwd"], fl, fl, net, pack, N, g_sdf, g_grad, da1v, *a, nbytes=by, **kw)

        # HBM-shaped families (no contraction): algorithmic bytes only
        def enc_fwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, Y, T=None):
            P = x.shape[0]
            by = P * (12.0 + geom.n_levels * 8 * 2 * 4 + 4.0 * Y.shape[1] * (4 if T is not None else 1))
            return t._timed("encode_fwd_kernel", o["encode_fwd"], 0.0, 0.0, geom, table, x, mode, include_x, pe_freqs, pe_max_exp, Y, T, nbytes=by)

        def enc_bwd(geom, table, x, mode, include_x, pe_freqs, pe_max_exp, dY, dT, dtable, dx=None, **kw):
            P = x.shape[0]
            # the gradient rows in, one read-modify-write of 8 corners x 2 floats per level and point (1 024 B / point each way)
            by = P * (12.0 + 4.0 * dY.shape[1] * (4 if dT is not None else 1) + (geom.n_levels * 8 * 2 * 4 * 2 if dtable is not None else 0))
            return t._timed("encode_bwd (owner scatter + bitmaps + pack)", o["encode_bwd"], 0.0, 0.0, geom, table, x, mode, include_x, pe_freqs,
                            pe_max_exp, dY, dT, dtable, dx, nbytes=by, **kw)

        def sdf_fwd(net, stream, table, E, M, a0, a1, sdf):
            fl = 2.0 * M * (net.in_dim * 256 + 256 * 256 + 256)
            by = 4.0 * M * (E.shape[1] + (512 if a0 is not None else 0) + 1)
            return t._timed("sdf_fwd_kernel", o["sdf_chain_fwd"], fl, fl, net, stream, table, E, M, a0, a1, sdf, nbytes=by)

        def sdf_bwd(net, stream, table, M, g_sdf, a0, a1, dz1, dz0, dE, *a, **kw):
            fl = 2.0 * M * (256 * 256 + net.in_dim * 256 + 256)
            by = 4.0 * M * (1 + 512 + 512 + (dE.shape[1] if dE is not None else 0))
            return t._timed("sdf_bwd_kernel", o["sdf_chain_bwd"], fl, fl, net, stream, table, M, g_sdf, a0, a1, dz1, dz0, dE, *a, nbytes=by, **kw)

        def colsum(X, nt, rows, out, *a, **kw):
            return t._timed("native_weighted_colsum_kernel", o["native_weighted_colsum"], 0.0, 0.0, X, nt, rows, out, *a,
                            nbytes=4.0 * rows * (32 * nt + 4), **kw)

        def adam(p, g, m, v, *a, **kw):
            return t._timed("adam_kernel", o["adam_step"], 0.0, 0.0, p, g, m, v, *a, nbytes=28.0 * p.numel(), **kw)  # p, m, v read + written, g read

        hip.encode_fwd, hip.encode_bwd, hip.sdf_chain_fwd, hip.sdf_chain_bwd, hip.native_weighted_colsum, hip.adam_step = (
            enc_fwd, enc_bwd, sdf_fwd, sdf_bwd, colsum, adam)
        hip.gemm, hip.gemm_planes, hip.film_chain_fwd, hip.film_chain_bwd_film, hip.film_chain_bwd_map = gemm, gemm_planes, fwd, bwd_film, bwd_map
        hip.wgrad_native_batch = wgrad
        hip.field_geo_fwd, hip.field_colour_fwd, hip.field_colour_bwd, hip.field_geo_bwd = geo_fwd, col_fwd, col_bwd, geo_bwd

    def install_attention(self):
        """the attention core of the RENI++ decoder (csrc/attention.hip): per (row, head) two L x 48 products forward (scores, values),
        five backward (scores, dP, dQ~, dK~, dV~); bytes: the row matrices in and out once, the per-camera K~ / V~ (and their gradients)"""
        from neusky_amd import hip
        names = ("attn_core_fwd", "attn_core_bwd", "attn_core_rays_fwd", "attn_core_rays_bwd")
        self._orig.update({n: getattr(hip, n) for n in names})
        o, t = self._orig, self

        def fwd(Q, dirs, Kt, Vt, scale, O, rmax, rsum):
            U, D, H = Q.shape
            nh, L = Kt.shape[1], Kt.shape[2]
            fl = 2.0 * U * D * nh * L * 48 * 2
            by = 4.0 * (2 * U * D * H + 2 * U * nh * L * 48 + 2 * U * nh * D)
            return t._timed("attn_core_fwd (matrix-core grid kernel)", o["attn_core_fwd"], fl, fl, Q, dirs, Kt, Vt, scale, O, rmax, rsum, nbytes=by)

        def bwd(Q, dirs, Kt, Vt, O, rmax, rsum, dO, scale, dQ, dKt, dVt):
            U, D, H = Q.shape
            nh, L = Kt.shape[1], Kt.shape[2]
            fl = 2.0 * U * D * nh * L * 48 * 5
            by = 4.0 * (4 * U * D * H + 4 * U * nh * L * 48 + 2 * U * nh * D)
            return t._timed("attn_core_bwd (rows + tokens kernels)", o["attn_core_bwd"], fl, fl, Q, dirs, Kt, Vt, O, rmax, rsum, dO, scale, dQ, dKt, dVt, nbytes=by)

        def rays_fwd(Q, *a, **kw):
            return t._timed("attn_core_rays_fwd", o["attn_core_rays_fwd"], 0.0, 0.0, Q, *a, nbytes=4.0 * 2 * Q.numel(), **kw)

        def rays_bwd(Q, *a, **kw):
            return t._timed("attn_core_rays_bwd", o["attn_core_rays_bwd"], 0.0, 0.0, Q, *a, nbytes=4.0 * 4 * Q.numel(), **kw)

        hip.attn_core_fwd, hip.attn_core_bwd, hip.attn_core_rays_fwd, hip.attn_core_rays_bwd = fwd, bwd, rays_fwd, rays_bwd

    def uninstall(self):
        from neusky_amd import hip
        for n, f in self._orig.items():
            setattr(hip, n, f)

    def summary(self):
        out = []
        for key, recs in self.records.items():
            # per-launch times grouped by the launch's size; a family's time = sum over its distinct launches of the MEDIAN over the
            # timing iterations (the first eager iteration after the graph replays allocates fresh blocks: a kernel that first touches
            # gigabytes of newly mapped memory was seen at 21 ms against 2.4)
            by_size = {}
            for r in recs:
                by_size.setdefault((r[2], r[4]), []).append(r[0].elapsed_time(r[1]))
            med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
            ms = sum(med(v) * len(v) for v in by_size.values())
            d = {"kernel": key, "launches": len(recs), "total_ms": ms, "avg_launch_ms": ms / len(recs),
                 "launch_ms_min_max": [min(min(v) for v in by_size.values()), max(max(v) for v in by_size.values())],
                 "algorithmic_flops_per_launch": sum(r[2] for r in recs) / len(recs),
                 "executed_flops_per_launch": sum(r[3] for r in recs) / len(recs),
                 "achieved_tflops": sum(r[2] for r in recs) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0}
            if all(r[4] is not None for r in recs):  # a streaming kernel: its HBM roofline beside the MFMA one
                d["algorithmic_bytes_per_launch"] = sum(r[4] for r in recs) / len(recs)
                d["achieved_GBs"] = sum(r[4] for r in recs) / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out.append(d)
        return sorted(out, key=lambda d: -d["total_ms"])


def _timed_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def forward_only_line(pipe, device):
    """BASELINE configs[1]: NeuSky SDF + albedo forward only, 1024 rays x 96 samples, hash L=16 F=2, no autograd: the whole field pass
    (hash encode with tangents -> fused geometry net + SDF normals -> colour net) and its hash-encode kernel alone against the HBM
    roofline (algorithmic bytes per point: position + 16 levels x 8 corners x 2 x 4 B gathered + the rows written)."""
    from neusky_amd import hip
    field = pipe.model.field
    P = RAYS * SAMPLES
    g = torch.Generator().manual_seed(0)
    x = ((torch.rand(P, 3, generator=g) * 2 - 1) * 0.8).to(device)
    with torch.no_grad():
        pipe.model.begin_step()
        whole = _timed_ms(lambda: field.field_values(x, want_albedo=True))
        geom, table = field.geom, field.encoding.table
        ldy = (3 + 6 * 6 + geom.n_levels * 2 + 3) // 4 * 4
        Y, T = torch.empty(P, ldy, device=device), torch.empty(3, P, ldy, device=device)
        enc_t = _timed_ms(lambda: hip.encode_fwd(geom, table, x, field.grid_mode, True, 6, 5.0, Y, T))
    nbytes = P * (12 + geom.n_levels * 8 * 2 * 4 + 4 * ldy * 4)
    return {"workload": "SDF+albedo forward only, 1024 rays x 96 samples, hash L=16 F=2 (BASELINE configs[1])", "ms": whole,
            "rays_per_s": RAYS / whole * 1e3, "points_per_s": P / whole * 1e3, "cpu_baseline": forward_only_cpu(pipe),
            "encode_fwd": {"ms": enc_t, "algorithmic_bytes": nbytes, "achieved_GBs": nbytes / enc_t / 1e6, "peak_GBs": HBM_PEAK_GBS,
                           "frac": nbytes / enc_t / 1e6 / HBM_PEAK_GBS, "bound": "hbm (the 48.8 MB table is Infinity-Cache resident)"}}


def forward_only_cpu(pipe, rays=64):
    """the CPU leg of configs[1] (SURVEY 8(d) / BASELINE.md section 5): the oracle's field pass (hash encode -> geometry net with the
    autograd normal -> colour net -> NeuS alpha, torch-CPU fp32, all host cores) on a bounded sample of the same workload: `rays` rays x
    96 samples with the full-size tables and networks; median of 5"""
    from oracle import neusky_oracle as O
    from util_step import oracle_params, oracle_step_cfg
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    p = oracle_params(pipe, dtype=torch.float32)
    cfg = oracle_step_cfg(pipe)
    g = torch.Generator().manual_seed(0)
    o = (torch.rand(rays, 3, generator=g) * 2 - 1) * 0.3
    d = torch.nn.functional.normalize(torch.randn(rays, 3, generator=g), dim=-1)
    eb = torch.linspace(0.05, 1.5, SAMPLES + 1)[None].expand(rays, -1).contiguous()

    def once():
        t0 = time.perf_counter()
        O.field_pass(p, cfg, o, d, eb)
        return time.perf_counter() - t0

    once()
    ts = sorted(once() for _ in range(5))
    dt = ts[2]
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "kind": "port",
            "sample": f"median of 5 field passes (oracle.field_pass, torch-CPU fp32) of {rays} rays x {SAMPLES} samples, full-size hash table and networks; {dt * 1e3:.0f} ms/pass"}


def envmap_decode_line(device):
    """BASELINE configs[0]: RENI++-shaped env-map decode, latent 36 x 3, 64 x 128 equirectangular directions -- the reference's own
    CPU-runnable case (neusky_model.py:351,1257-1271): the torch-CPU oracle (float32, `oracle.reni_decode`: illumination-prior plumbing,
    no GPU) timed beside the HIP decoder on the same latent; both the median of 10 decodes; their maps must agree (tests/test_gpu_reni_envmap.py)."""
    import math
    from oracle import neusky_oracle as O
    from neusky_amd.model_components.illumination import RENIField, RENIFieldConfig
    torch.manual_seed(0)
    field = RENIField(RENIFieldConfig(latent_dim=36)).to(device)
    h, w = 64, 128
    phi = (torch.arange(h, dtype=torch.float32) + 0.5) / h * math.pi
    theta = (torch.arange(w, dtype=torch.float32) + 0.5) / w * 2 * math.pi
    P, T_ = torch.meshgrid(phi, theta, indexing="ij")
    dirs = torch.stack([torch.sin(P) * torch.cos(T_), torch.sin(P) * torch.sin(T_), torch.cos(P)], -1).reshape(-1, 3)
    Z = torch.randn(36, 3) * 0.4
    scale = torch.tensor(1.3)
    net = field.network
    lins = net.mapping_network.linears()
    c = lambda t: t.detach().cpu().float()  # noqa: E731
    prm = {}
    for i, lin in enumerate(lins[:-1]):
        prm[f"reni.map_w{i}"], prm[f"reni.map_b{i}"] = c(lin.weight), c(lin.bias)
    prm["reni.map_wo"], prm["reni.map_bo"] = c(lins[-1].weight), c(lins[-1].bias)
    for i, l in enumerate(net.net):
        prm[f"reni.film_w{i}"], prm[f"reni.film_b{i}"] = c(l.layer.weight), c(l.layer.bias)
    prm["reni.out_w"], prm["reni.out_b"] = c(net.final_layer.weight), c(net.final_layer.bias)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)

    def cpu_once():
        t0 = time.perf_counter()
        with torch.no_grad():
            out = O.reni_decode(Z[None].expand(dirs.shape[0], -1, -1), dirs, scale.expand(dirs.shape[0]), prm)
        return (time.perf_counter() - t0) * 1e3, out

    cpu_once()
    cpu = sorted(cpu_once()[0] for _ in range(10))
    ref = cpu_once()[1]
    dd, zd, sd = dirs.to(device), Z.to(device)[None], scale.to(device)[None]
    with torch.no_grad():
        got = field.forward_grid(dd, zd, sd)[0]
        gpu = sorted(_timed_ms(lambda: field.forward_grid(dd, zd, sd), iters=1, warm=1) for _ in range(10))
    rel = float((got.cpu() - ref).abs().max() / ref.abs().max())
    return {"workload": "RENI++-shaped env-map decode, latent 36 x 3, 64 x 128 equirect = 8192 directions (BASELINE configs[0])",
            "cpu_ms": cpu[len(cpu) // 2], "cpu_threads": cores, "cpu_kind": "port (oracle.reni_decode, torch-CPU fp32; the reference's decoder source is absent: parity unpinned)",
            "hip_ms": gpu[len(gpu) // 2], "max_rel_diff_hip_vs_cpu": rel, "median_of": 10}


def attention_decoder_line(device, steps=5):
    """the SAME training step with the illumination decoder the reference configures (neusky_config.py:78-95: conditioning="Attention",
    VN / SO2-about-z invariance, 8 heads x 6 layers, hidden 128) in place of the FiLM-SIREN decoder of the headline: this project's
    restatement of the published RENI++ architecture (model_components/illumination.py:AttentionDecoder; ns_reni's source and weights
    are absent from the reference tree: PARITY UNPINNED, own oracle oracle.reni_attention_decode).  HIP-graph replay like the headline,
    median of `steps` iterations."""
    from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
    from neusky_amd.configs.neusky_config import synthetic_pipeline_config
    from neusky_amd.utils.randomise import randomise
    cfg = synthetic_pipeline_config()
    cfg.model.num_neus_samples_per_ray = SAMPLES
    cfg.model.num_proposal_samples_per_ray = tuple(PROPOSAL)
    cfg.model.illumination_sampler.num_directions = DIRECTIONS
    cfg.datamanager.train_num_rays_per_batch = RAYS
    cfg.model.illumination_field.conditioning = "Attention"
    torch.manual_seed(0)
    pipe = cfg.setup(device=device, world_size=1, local_rank=0)
    pipe.train()
    randomise(pipe)
    from neusky_amd.engine import GraphedTrainStep
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    batches = [pipe.datamanager.next_train(i) for i in range(3)]
    launch = "HIP graph replay"
    try:
        stepper = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=3000)
        run = lambda i: stepper.step(3002 + i, batches[i % 3][0], batches[i % 3][1])[0]  # noqa: E731
    except Exception as e:  # capture failed: fall back to host launches (and say so)
        launch = f"eager (graph capture failed: {type(e).__name__})"
        run = lambda i: train_iteration(pipe, opt, 3002 + i, ray_bundle=batches[i % 3][0], batch=batches[i % 3][1])[0]  # noqa: E731
    for i in range(2):
        run(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        loss = run(2 + i)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    ms = ts[len(ts) // 2]
    # the decoder's own kernels, timed like the headline's: HIP events around every launch of two eager iterations on one stream
    import neusky_amd.ops as ops
    timer = KernelTimer()
    timer.install()
    timer.install_attention()
    keep = (pipe.model.second_stream, ops.ASYNC_WGRAD)
    pipe.model.second_stream, ops.ASYNC_WGRAD = False, False
    try:
        for i in range(2):
            train_iteration(pipe, opt, 3100 + i, ray_bundle=batches[i][0], batch=batches[i][1])
        torch.cuda.synchronize()
    finally:
        pipe.model.second_stream, ops.ASYNC_WGRAD = keep
        timer.uninstall()
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    fams = []
    for k in timer.summary():
        if not (k["kernel"].startswith("attn_") or k["kernel"].startswith("gemm_")):
            continue  # (the rest of the step is the headline's: its families are in the headline's `kernels`)
        fams.append({"kernel": k["kernel"], "ms_per_step": k["total_ms"] / 2, "launches_per_step": k["launches"] / 2,
                     "mfma_frac_of_833_tflops": k["achieved_tflops"] / peak, "hbm_frac_of_8000_GBs": k["achieved_GBs"] / HBM_PEAK_GBS if "achieved_GBs" in k else None})
    out = {"workload": "full NeuSky train step as the headline, illumination decoder = RENI++ attention decoder (neusky_config.py:78-95): "
                       "300 cameras x 512 directions + 1024 ray rows, 100 tokens, 8 heads x 6 layers, hidden 128; attention core on "
                       "csrc/attention.hip (matrix-core forms for the camera grids, per-camera ray kernels for the rays' own rows), residual add + layer norm "
                       "fused (same file), linear layers on this package's dense-layer kernels",
           "ms_per_step": ms, "rays_per_s": RAYS / (ms * 1e-3), "launch": launch, "final_loss": float(loss),
           "parity": "unpinned (decoder source and weights absent from the reference tree); HIP path vs oracle.reni_attention_decode: tests/test_illumination_attention.py",
           "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
           "roofline": {"note": "the decoder's own kernel families (attention core, dense layers: exact / bf16 x 3 / fp16-split GEMMs), HIP events around every "
                                "launch of two eager single-stream iterations; fractions of 833 TFLOP/s (algorithmic FLOPs) and of 8 TB/s (operands once)",
                        "dominant": fams[0] if fams else None, "families": fams[:6]}}
    del pipe, opt
    torch.cuda.empty_cache()
    return out


def trainer_surface_line(device, steps=10):
    """The same step driven the way nerfstudio's Trainer drives the plugin surface (neusky_pipeline.py:198-200,
    241-291) -- zero_grad -> pipeline.get_train_loss_dict -> sum the loss dict -> backward -> optimizer.step per group ->
    scheduler.step -- with nothing from neusky_amd.engine.  Two optimizers: `neusky_amd.optimizers.SlabAdam`, what the plugin's
    `SlabAdamOptimizerConfig` hands the trainer (torch.optim.Adam's update, one fused launch per group: `ms_per_step`), and
    torch.optim.Adam(eps=1e-15) itself (`torch_adam_ms_per_step`); each with the pipeline's graph replay
    (NeuSkyPipelineConfig.graph_replay) and without it (`eager_*`: host launches of every kernel); inputs resident in HBM like the
    headline's; `with_datamanager_ms_per_step`: get_train_loss_dict(step) drawing its own batch and sky rays each step, as the Trainer
    calls it.  Fresh pipelines at full size."""
    import functools
    from neusky_amd.engine import ExponentialDecaySchedulerConfig, neusky_optimizers
    from neusky_amd.optimizers import SlabAdam
    from neusky_amd.utils.randomise import randomise

    def run(fused):
        torch.manual_seed(0)
        pipe = build_pipeline(device, 1, 0)
        randomise(pipe)
        cfg = neusky_optimizers()
        opts, scheds = {}, {}
        Adam = SlabAdam if fused else torch.optim.Adam
        for name, params in pipe.get_param_groups().items():
            oc, sc = cfg[name]["optimizer"], cfg[name]["scheduler"]
            opts[name] = Adam([p for p in params if p.requires_grad], lr=oc.lr, eps=oc.eps, betas=oc.betas)
            if isinstance(sc, ExponentialDecaySchedulerConfig):
                sc.lr_init = oc.lr
            scheds[name] = torch.optim.lr_scheduler.LambdaLR(opts[name], lr_lambda=lambda e, f=sc.factor: f(1000 + e))
        batches = [pipe.datamanager.next_train(i) for i in range(3)]
        skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(3)]

        def iteration(step, resident=True):
            for o in opts.values():
                o.zero_grad()
            j = step % 3
            kw = dict(ray_bundle=batches[j][0], batch=batches[j][1], randoms={"sky_ray_bundle": skies[j]}) if resident else {}
            _, loss_dict, _ = pipe.get_train_loss_dict(step, **kw)
            loss = functools.reduce(torch.add, loss_dict.values())
            loss.backward()
            for o in opts.values():
                o.step()
            for s_ in scheds.values():
                s_.step()
            return loss

        def timed(n, resident=True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                loss = iteration(2000 + i, resident)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, float(loss)

        for i in range(3):  # eager: also the warm-up in front of the capture
            iteration(1000 + i)
        eager_ms, _ = timed(max(steps // 2, 3))
        pipe.config.graph_replay, pipe.config.graph_replay_warmup = True, 0
        for i in range(3):  # the first call captures
            iteration(1500 + i)
        assert pipe._train_graph is not None
        ms, final = timed(steps)
        dm_ms, _ = timed(steps, resident=False)
        del pipe, opts, scheds
        torch.cuda.empty_cache()
        return ms, eager_ms, dm_ms, final

    ms, eager_ms, dm_ms, final = run(True)
    t_ms, t_eager_ms, _, _ = run(False)
    return {"loop": "zero_grad -> get_train_loss_dict -> reduce(add, loss_dict.values()) -> backward -> optimizer.step x 5 groups -> LambdaLR.step",
            "optimizer": "neusky_amd.optimizers.SlabAdam (plugin.SlabAdamOptimizerConfig: torch.optim.Adam's update, one nsky_adam_step launch per group, gradients read in place from the pipeline's slab)",
            "ms_per_step": ms, "rays_per_s": RAYS / (ms * 1e-3), "eager_ms_per_step": eager_ms, "with_datamanager_ms_per_step": dm_ms,
            "torch_adam_ms_per_step": t_ms, "torch_adam_eager_ms_per_step": t_eager_ms, "final_loss": final,
            "launch": "pipeline graph replay (NeuSkyPipelineConfig.graph_replay) + the optimizers' own launches"}


def global_batch_line(device, rays=8192, steps=5):
    """BASELINE configs[3]'s GLOBAL batch (8192 rays) as ONE GPU's batch (eager launches, median of `steps`): what the 288 GB of one
    MI355X allow when no 8-GPU node is at hand -- 2.1 M DDF rows per step; fixed per-step costs (slab fills, Adam over the hash tables,
    weight packing) and the last-round tails of the big launches are amortised over 8 x the rays.  Not the headline (that is 1024
    rays per GPU, weak scaling); parity at this size: tests/test_gpu_full_size.py::test_global_batch_of_configs3_on_one_gpu_forward_slice."""
    from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
    from neusky_amd.utils.randomise import randomise
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    pipe = build_pipeline(device, 1, 0, rays=rays)
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    batches = [pipe.datamanager.next_train(i) for i in range(3)]
    for i in range(2):
        train_iteration(pipe, opt, 3000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
    torch.cuda.synchronize()
    ts = []
    for i in range(steps):
        t0 = time.perf_counter()
        train_iteration(pipe, opt, 3002 + i, ray_bundle=batches[i % 3][0], batch=batches[i % 3][1])
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    ms = ts[len(ts) // 2]
    out = {"workload": f"full NeuSky train step, {rays} rays x {SAMPLES} samples x {DIRECTIONS} directions on ONE GPU (the global batch of BASELINE configs[3])",
           "ms_per_step": ms, "rays_per_s": rays / (ms * 1e-3), "launch": "eager", "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9}
    del pipe, opt
    torch.cuda.empty_cache()
    return out


def frame_1080p_rays(pipe, device, H=1080, W=1920):
    """the 1920 x 1080 pinhole frame of the render-pass configuration: camera 0 of the synthetic scene, focal 1100 px
    (-> the frame's ray bundle, a bundle factory for sub-frames, the camera position and the [H, W, 3] unit directions)"""
    from neusky_amd.cameras.rays import RayBundle
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d_cam = torch.stack([(xs - W / 2) / 1100.0, (ys - H / 2) / 1100.0, torch.ones(H, W)], -1)
    cR, cp = pipe.datamanager.cam_R[0], pipe.datamanager.cam_pos[0]
    d = torch.einsum("ij,hwj->hwi", cR, d_cam)
    d = d / d.norm(dim=-1, keepdim=True)
    mk = lambda o, dd, h, w: RayBundle(origins=o.to(device), directions=dd.to(device), pixel_area=torch.ones(h, w, 1, device=device),  # noqa: E731
                                       camera_indices=torch.zeros(h, w, 1, dtype=torch.long, device=device),
                                       metadata={"directions_norm": torch.ones(h, w, 1, device=device)})
    return mk(cp.expand(H, W, 3).contiguous(), d, H, W), mk, cp, d


def render_1080p_line(pipe, device, chunk=4096):
    """BASELINE configs[4]: relighting render pass, one 1920 x 1080 frame, 512 illumination directions = 256 upper-hemisphere DDF
    visibility queries per ray, static chunks replayed from a HIP graph (publication/render_animation.py:118-119,196-207;
    neusky_model.py:1369-1501).  ms_per_frame: the second frame (the chunk's graph exists); first_frame_ms includes its capture; the
    dominant kernel's MFMA fraction comes from HIP events around one eagerly launched chunk."""
    H, W = 1080, 1920
    pipe.eval()
    rb, mk, cp, d = frame_1080p_rays(pipe, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=0, chunk=chunk, use_graph=True)
    torch.cuda.synchronize()
    dt_first = time.perf_counter() - t0  # with the graph capture of the chunk (an animation pays it once: render_animation.py renders hundreds of frames)
    t0 = time.perf_counter()
    out = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=0, chunk=chunk, use_graph=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer = KernelTimer()
    timer.install()
    try:
        rows = chunk // W + 1
        small = mk(cp.expand(rows, W, 3).contiguous(), d[:rows].contiguous(), rows, W)
        pipe.model.get_outputs_for_camera_ray_bundle(small, camera_index=0, chunk=chunk, use_graph=False)
        torch.cuda.synchronize()
    finally:
        timer.uninstall()
    ks = timer.summary()
    dom = ks[0] if ks else None
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    res = {"workload": "relighting render pass, 1920x1080, 512 directions (256 DDF visibility queries/ray), HIP-graph-replayed chunks (BASELINE configs[4])",
           "ms_per_frame": dt * 1e3, "first_frame_ms": dt_first * 1e3, "rays_per_s": H * W / dt, "chunk_rays": chunk,
           "rgb_mean": float(out["rgb"].mean())}
    if dom is not None:
        res["dominant_kernel"] = {"kernel": dom["kernel"], "avg_launch_ms": dom["avg_launch_ms"], "achieved_tflops": dom["achieved_tflops"],
                                  "frac_of_833_tflops": dom["achieved_tflops"] / peak}
    pipe.train()
    return res


def render_pmc_traffic(timeout_s: float = 200.0):
    """HBM traffic of the render pass from two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE) of ONE eagerly launched
    480 x 270 frame (tools/bench_render.py 270 480 eager: same chunk size and kernels as the 1080p frame, 1/16 of its rays, one dispatch
    record per launch) -> (dict, note) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    base = tempfile.mkdtemp(prefix="nsky_pmc_render_")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NSKY_BENCH_LAUNCHER"):
        env.pop(k, None)
    try:
        dirs = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            dirs[c] = os.path.join(base, c)
            cmd = [exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", dirs[c], "--", sys.executable,
                   os.path.join(ROOT, "tools", "bench_render.py"), "270", "480", "eager"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {c} pass of the render exited with code {r.returncode}"
        t = pmc_traffic(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"], iters=1.0)
        return {"frame": "480x270, eager launches, chunk 4096", "GB_per_480x270_frame": t["whole_step_GB"],
                "GB_per_1080p_frame_scaled_x16": 16.0 * t["whole_step_GB"], "bytes_per_launch": t["bytes_per_launch"],
                "GB_per_frame_by_kernel": t["GB_per_iteration"]}, "collected in this run"
    except Exception as exc:  # noqa: BLE001
        return None, f"{type(exc).__name__}: {str(exc)[:160]}"
    finally:
        shutil.rmtree(base, ignore_errors=True)


STEP_KERNEL_SOURCES_EXCLUDE = ("attention.hip",)  # kernels of the attention-decoder key only: not in the headline step the counters describe


def step_kernel_sources():
    csrc = os.path.join(ROOT, "neusky_amd", "csrc")
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f not in STEP_KERNEL_SOURCES_EXCLUDE]


def _sources_sha(paths) -> str:
    import hashlib
    h = hashlib.sha256()
    for f in sorted(paths):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(fetch_dir: str, write_dir: str, iters: float = 0.0) -> dict:
    """HBM traffic by kernel family from two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) of an
    eager bench run: bytes = 1024 * (2 * FETCH_SIZE + WRITE_SIZE) -- the counters are KiB, and gfx950 reports half the bytes of wide
    coalesced reads (MI355X_MICROARCH.md).  -> {whole_step_GB, iterations, bytes_per_launch{family}, GB_per_iteration{family}}"""
    import collections
    import csv
    import glob
    import re
    tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for c, idx, d in (("FETCH_SIZE", 0, fetch_dir), ("WRITE_SIZE", 1, write_dir)):
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        for f in sorted(files, key=os.path.getmtime)[-1:]:  # the newest pass only
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c:
                    k = re.sub(r"\(anonymous namespace\)::|^void ", "", row["Kernel_Name"])
                    k = re.split(r"\((?![a-z])", k)[0][:90]
                    tot[k][idx] += float(row["Counter_Value"]) * 1024.0
                    if idx == 0:
                        tot[k][2] += 1
    if not tot:
        raise RuntimeError("no counter records found")
    if iters <= 0:  # one hemisphere-composite launch per training iteration
        iters = float(max(v[2] for k, v in tot.items() if k.startswith("hemi_fwd_kernel")))
    fam = collections.defaultdict(lambda: [0.0, 0])
    for k, v in tot.items():
        name = k.split("(")[0].strip()
        key = re.sub(r"<(\d+)[^>]*>", r"<\1>", name) if name.startswith("film_") else name.split("<")[0]
        fam[key][0] += 2 * v[0] + v[1]
        fam[key][1] += v[2]
    total = sum(v[0] for v in fam.values()) / iters
    return {"whole_step_GB": total / 1e9, "iterations": iters,
            "fetch_GB": sum(2 * v[0] for v in tot.values()) / iters / 1e9, "write_GB": sum(v[1] for v in tot.values()) / iters / 1e9,
            "bytes_per_launch": {k: v[0] / max(v[1], 1) for k, v in fam.items() if v[1] > 0 and v[0] / iters > 5e7},
            "launches_per_iteration": {k: v[1] / iters for k, v in fam.items() if v[1] > 0 and v[0] / iters > 5e7},
            "GB_per_iteration": {k: v[0] / iters / 1e9 for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:20]}}


def _family_of(kernel_name: str) -> str:
    import re
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", kernel_name)
    k = re.split(r"\((?![a-z])", k)[0][:90]
    name = k.split("(")[0].strip()
    return re.sub(r"<(\d+)[^>]*>", r"<\1>", name) if name.startswith("film_") else name.split("<")[0]


def pmc_mfma(sq_dir: str) -> dict:
    """matrix-pipe occupancy and effective shader clock by kernel family from ONE rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES
    GRBM_GUI_ACTIVE pass of an eager bench run.  MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (32 per
    32x32x16 MFMA), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so a dispatch lasts GUI_ACTIVE / 8
    cycles:  mfma_busy = MFMA_BUSY / (1024 * GUI_ACTIVE / 8),  clock = GUI_ACTIVE / 8 / duration (reads high below ~0.3 ms)."""
    import collections
    import csv
    import glob
    acc = collections.defaultdict(lambda: {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "n": 0, "dur_ns": 0.0})
    cfiles = sorted(glob.glob(os.path.join(sq_dir, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    for f in cfiles:
        for row in csv.DictReader(open(f)):
            c = row["Counter_Name"]
            if c in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                a = acc[_family_of(row["Kernel_Name"])]
                a[c] += float(row["Counter_Value"])
                if c == "GRBM_GUI_ACTIVE":
                    a["n"] += 1
                    if "Start_Timestamp" in row and "End_Timestamp" in row:
                        a["dur_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    if not any(a["dur_ns"] for a in acc.values()):  # older layouts: the durations sit in the kernel trace of the same pass
        for f in sorted(glob.glob(os.path.join(sq_dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for row in csv.DictReader(open(f)):
                fam = _family_of(row["Kernel_Name"])
                if fam in acc:
                    acc[fam]["dur_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    out = {}
    for fam, a in acc.items():
        if a["GRBM_GUI_ACTIVE"] > 0 and a["n"] > 0:
            cyc = a["GRBM_GUI_ACTIVE"] / 8.0
            out[fam] = {"mfma_busy": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), "launches": a["n"],
                        "avg_launch_us_under_counters": a["dur_ns"] / a["n"] * 1e-3 if a["dur_ns"] else None,
                        "clock_mhz": cyc / (a["dur_ns"] * 1e-9) * 1e-6 if a["dur_ns"] else None}
    if not out:
        raise RuntimeError("no SQ counter records found")
    return out


class PowerSampler:
    """package power (and the shader clock level, where the driver exposes it) of the GPU during the timed region, from sysfs
    (hwmon power1_average / power1_input in microwatts, pp_dpm_sclk's starred level), ~50 samples per second on a host thread; an
    ordinary user may read them.  Anything missing -> the fields stay None."""

    def __init__(self, index: int = 0):
        import glob
        self.samples, self.clocks, self._stop, self._th = [], [], False, None
        self.power_cap_w = None
        self.power_files, self.sclk_files = [], []
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
        cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk")) or glob.glob(os.path.join(c, "hwmon/hwmon*/power1_*"))]
        # the box may expose more cards in sysfs than the one this process was given: match the PCI address of torch's device
        self.card = None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
            for c in cards:
                if os.path.basename(os.path.realpath(c)).lower().startswith(want):
                    self.card = c
        except Exception:  # noqa: BLE001  (an older torch without the PCI fields)
            pass
        if self.card is None and len(cards) == 1:
            self.card = cards[0]
        if self.card is not None:
            c = self.card
            self.power_files = [f for f in (glob.glob(os.path.join(c, "hwmon/hwmon*/power1_average")) + glob.glob(os.path.join(c, "hwmon/hwmon*/power1_input")))][:1]
            self.sclk_files = [os.path.join(c, "pp_dpm_sclk")] if os.path.exists(os.path.join(c, "pp_dpm_sclk")) else []
            for f in glob.glob(os.path.join(c, "hwmon/hwmon*/power1_cap")):  # the package power limit in force on this box (microwatts)
                try:
                    self.power_cap_w = float(open(f).read().strip()) * 1e-6
                except (OSError, ValueError):
                    pass

    def _run(self):
        import re
        while not self._stop:
            try:
                for f in self.power_files:
                    self.samples.append(float(open(f).read().strip()) * 1e-6)
                for f in self.sclk_files:
                    m = re.search(r"(\d+)\s*Mhz\s*\*", open(f).read(), re.I)
                    if m:
                        self.clocks.append(float(m.group(1)))
            except (OSError, ValueError):
                pass
            time.sleep(0.02)

    def __enter__(self):
        import threading
        if self.power_files or self.sclk_files:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=1.0)

    def summary(self):
        avg = lambda v: (sum(v) / len(v)) if v else None  # noqa: E731
        return {"power_w": avg(self.samples), "power_w_max": max(self.samples) if self.samples else None, "sclk_level_mhz": avg(self.clocks),
                "power_cap_w": self.power_cap_w, "samples": max(len(self.samples), len(self.clocks)), "card": self.card,
                "source": "sysfs hwmon power1_average / pp_dpm_sclk of the card with this device's PCI address, during the timed region"
                if (self.samples or self.clocks) else "not exposed to this user on this box"}


def live_pmc_traffic(timeout_s: float = 240.0):
    """the two --pmc passes collected IN THIS RUN: rank 0 (N = 1) starts `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py
    --no-spawn ...` (an eager two-step bench: one dispatch record per launch) as a child process, twice, and reads the counters back.
    Counters only with --kernel-trace, as the pool's rules ask.  Any failure (no rocprofv3, a timeout, no records) returns (None, reason):
    the line then falls back to the committed file and says so."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    base = tempfile.mkdtemp(prefix="nsky_pmc_")
    dirs = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "NSKY_BENCH_LAUNCHER"):
        env.pop(k, None)
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            dirs[c] = os.path.join(base, c)
            cmd = [exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", dirs[c], "--", sys.executable, os.path.abspath(__file__),
                   "--no-spawn", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph", "--no-exact-f32", "--no-extra-configs",
                   "--no-live-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {c} pass exited with code {r.returncode}"
        out = pmc_traffic(dirs["FETCH_SIZE"], dirs["WRITE_SIZE"])
        try:  # third pass: matrix-pipe busy cycles and the effective clock per kernel family
            dirs["SQ"] = os.path.join(base, "SQ")
            cmd = [exe, "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "--output-format", "csv", "-d", dirs["SQ"], "--",
                   sys.executable, os.path.abspath(__file__), "--no-spawn", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-graph",
                   "--no-exact-f32", "--no-extra-configs", "--no-live-pmc"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            out["mfma"] = pmc_mfma(dirs["SQ"]) if r.returncode == 0 else {"error": f"SQ pass exited with code {r.returncode}"}
        except Exception as exc:  # noqa: BLE001
            out["mfma"] = {"error": f"{type(exc).__name__}: {str(exc)[:160]}"}
        return out, "collected in this run (rocprofv3 --kernel-trace --pmc passes of an eager two-step bench: FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE)"
    except Exception as exc:  # noqa: BLE001
        return None, f"{type(exc).__name__}: {str(exc)[:160]}"
    finally:
        shutil.rmtree(base, ignore_errors=True)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (oracle/neusky_oracle.py, torch-CPU fp32, all host cores) timed on a bounded sample of the same
    workload: full train step (forward + all losses + backward through torch autograd, as the reference does) on a
    few rays with the full 96 samples / 512 directions / full-size networks and hash tables."""
    from oracle import neusky_oracle as O
    from util_step import make_randoms, oracle_params, oracle_randoms, oracle_step_cfg, randomise
    cores = min(os.cpu_count() or 1, 32)  # more threads only add synchronisation overhead on the small CPU sample
    torch.set_num_threads(cores)
    rays = 16
    pipe_cpu = None
    # parameters come from a CPU-resident copy of the product's initial state (host modules only; no HIP call)
    from neusky_amd.configs.neusky_config import synthetic_pipeline_config
    cfg = synthetic_pipeline_config()
    cfg.model.num_neus_samples_per_ray = SAMPLES
    cfg.model.illumination_sampler.num_directions = DIRECTIONS
    cfg.datamanager.train_num_rays_per_batch = rays
    cfg.visibility_train_sampler.num_samples_on_sphere = 1
    cfg.visibility_train_sampler.num_rays_per_sample = 16
    cfg.num_sky_rays = 8
    pipe = cfg.setup(device="cpu")
    pipe.train()
    randomise(pipe)
    rb, batch = pipe.datamanager.next_train(0)
    rnd = make_randoms(pipe, rays)
    p = oracle_params(pipe, dtype=torch.float32)
    light = pipe.model.illumination_sampler(rotation=rnd["light_rotation"]).float()
    scfg = oracle_step_cfg(pipe)
    orr = oracle_randoms(rnd, light, dtype=torch.float32)

    def one():
        ld, _ = O.neusky_train_step(p, scfg, rb.origins.float(), rb.directions.float(), rb.camera_indices.reshape(-1),
                                    batch["image"].float(), batch["mask"], orr, light)
        loss = sum(ld.values())
        keys = [k for k in p if not k.startswith("reni.")]
        torch.autograd.grad(loss, [p[k] for k in keys], allow_unused=True)

    t0 = time.time(); one(); warm = time.time() - t0
    reps = max(0, min(11, int(seconds_budget / max(warm, 1e-3)) - 1))  # the median of >= 10 steps where the budget allows
    if reps == 0:  # a single step already exceeds the budget: report it rather than blow the bench wall clock
        dt, reps = warm, 1
    else:
        ts = []
        for _ in range(reps):
            t0 = time.time()
            one()
            ts.append(time.time() - t0)
        ts.sort()
        dt = ts[len(ts) // 2]
    return {"value": rays / dt, "unit": "rays/s", "cores": cores, "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"median of {reps} full train steps (fwd+bwd via torch autograd, fp32) of {rays} rays x {SAMPLES} samples x "
                      f"{DIRECTIONS} directions + 16 DDF-fit + 8 sky rays; Adam excluded; {dt:.2f} s/step"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying the captured HIP graph")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the three extra eager steps under the exact-fp32 policy")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the forward-only (configs[1]) and 1080p render (configs[4]) lines")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not collect the HBM counters in this run (roofline.traffic then comes from the committed file)")
    ap.add_argument("--launcher-selftest", choices=("ok", "fail"), default=None,
                    help="CPU check of the launch path alone (tests/test_cpu_abi_and_host.py): the ranks started by spawn_ranks rendezvous over gloo, "
                         "all-reduce a known answer on host tensors and rank 0 prints one line; 'fail': the last rank exits non-zero instead")
    ap.add_argument("--no-spawn", action="store_true",
                    help="N = 1 only: run the rank in THIS process, without a process group (for rocprofv3 -- python3 bench.py ...: the "
                         "profiler's library initialises the GPU in the process it preloads into, which then must not start ranks)")
    return ap.parse_args(argv)


def spawn_ranks(args, argv) -> int:
    """`python3 bench.py --gpus N` without an external launcher: THIS process touches no GPU (importing torch does not initialise HIP;
    nothing below calls into torch.cuda) and starts one child process per rank -- the same file with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, which is what `python -m torch.distributed.run` does and what the reference's launcher does per GPU
    (nerfstudio's train.py spawns one process per device around neusky_pipeline.py:198-200).  Children are started with subprocess (fork +
    exec in the child: this process is never replaced), rank 0's stdout is relayed (its ONE JSON line), the other ranks' stdout goes to
    stderr; the exit code is non-zero if any rank's is, and a failing rank takes the others down (by PID) instead of leaving them in a
    collective that can never complete."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NSKY_BENCH_LAUNCHER="self-spawned")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE, text=True))
    outs = [[] for _ in procs]

    def pump(i):
        for line in procs[i].stdout:
            outs[i].append(line)
            if i != 0:
                sys.stderr.write(f"[rank {i}] {line}")

    threads = [threading.Thread(target=pump, args=(i,), daemon=True) for i in range(n)]
    for t in threads:
        t.start()
    rc = 0
    live = set(range(n))
    while live:
        time.sleep(0.2)
        for i in sorted(live):
            code = procs[i].poll()
            if code is None:
                continue
            live.discard(i)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {i} exited with code {code}; stopping the other ranks\n")
                for j in sorted(live):
                    procs[j].terminate()  # exact PIDs this process started
    for t in threads:
        t.join(timeout=10)
    sys.stdout.write("".join(outs[0]))
    sys.stdout.flush()
    return rc


def rccl_selfcheck(opt, pipe, world, rank, device, backend):
    """the collectives of the N > 1 step really executed on this process group before the timed region, at every N (N = 1 included: a
    one-rank RCCL communicator): broadcast of the replicated state (neusky_pipeline.py:198-199), an all-reduce with a known answer, and
    the gradient slab's all-reduce (mean) timed -- the message of engine.Optimizers.all_reduce_gradients."""
    res = {"backend": backend, "ranks": dist.get_world_size()}
    on_dev = backend == "nccl"
    x = torch.full((1024,), float(rank + 1), device=device if on_dev else "cpu")
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
    res["all_reduce_known_answer"] = bool((x == world * (world + 1) / 2).all().item())
    res["tensors_broadcast"] = pipe.grad_sync.broadcast_parameters() if hasattr(pipe, "grad_sync") and pipe.grad_sync is not None else 0
    if on_dev:
        slab = opt.flat_g
        slab.zero_()
        for _ in range(2):
            dist.all_reduce(slab, op=dist.ReduceOp.AVG)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(slab, op=dist.ReduceOp.AVG)
        torch.cuda.synchronize()
        res["gradient_slab_bytes"] = slab.numel() * 4
        res["gradient_slab_all_reduce_ms"] = (time.perf_counter() - t0) / 5 * 1e3
        res["slab_stays_zero"] = bool((slab == 0).all().item())
    return res


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ  # torch.distributed.run, or a child of spawn_ranks
    if not launched and not args.no_spawn:
        # before ANY GPU call: this process only starts the ranks and relays rank 0's line
        raise SystemExit(spawn_ranks(args, argv))
    if args.no_spawn and args.gpus > 1:
        raise SystemExit("--no-spawn is the in-process N = 1 form (profilers); N > 1 starts one process per GPU")

    # ONE JSON line on stdout: libraries that print to file descriptor 1 (RCCL's version banner at the first collective) go to stderr;
    # the line itself is written to the saved descriptor
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.launcher_selftest is not None:  # no GPU, no pipeline: the environment spawn_ranks hands a rank, a real (gloo) process group
        world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
        assert int(os.environ["LOCAL_RANK"]) == rank and os.environ["MASTER_ADDR"] == "127.0.0.1" and world == args.gpus
        if args.launcher_selftest == "fail" and rank == world - 1:
            raise SystemExit(7)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        x = torch.full((8,), float(rank + 1))
        dist.all_reduce(x, op=dist.ReduceOp.SUM)
        if rank != 0:
            print(f"(a line of rank {rank}: must not reach the parent's stdout)")
        else:
            line_out.write(json.dumps({"selftest": "ok", "ranks": world, "sum": float(x[0]), "launcher": os.environ.get("NSKY_BENCH_LAUNCHER")}) + "\n")
            line_out.flush()
        dist.barrier()
        dist.destroy_process_group()
        return

    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if launched and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    launcher = os.environ.get("NSKY_BENCH_LAUNCHER", "external (torch.distributed.run)") if launched else "in-process (--no-spawn)"
    # (NSKY_BENCH_DEVICE / NSKY_DIST_BACKEND: the one-GPU test of this file's N > 1 path, tests/test_gpu_bench_two_ranks.py -- two ranks share
    # cuda:0 over gloo; a driver run never sets them)
    dev_index = int(os.environ.get("NSKY_BENCH_DEVICE", local_rank))
    backend = os.environ.get("NSKY_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"
    pg_note = None
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group(backend, rank=rank, world_size=world)  # backend "nccl" == RCCL on ROCm
        except Exception as exc:  # noqa: BLE001
            if world > 1:
                raise
            pg_note = f"one-rank process group refused: {type(exc).__name__}: {str(exc)[:160]}"  # N = 1 needs no collective: say so, go on

    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers, train_iteration
    torch.manual_seed(1234 + rank)
    pipe = build_pipeline(device, world, local_rank)
    from neusky_amd.utils.randomise import randomise  # package code: the oracle is only imported by the cpu_baseline leg
    randomise(pipe, seed=rank)  # identical replicas are restored by the parameter broadcast in the pipeline for N>1
    if world > 1:
        pipe.grad_sync.broadcast_parameters()
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups(), world_size=world)
    selfcheck = None
    if launched and dist.is_initialized():
        from neusky_amd.distributed import broadcast_module_state
        selfcheck = rccl_selfcheck(opt, pipe, world, rank, device, backend)
        if world == 1:  # (the pipeline only broadcasts at N > 1: here the one-rank broadcast is part of the check)
            selfcheck["tensors_broadcast"] = broadcast_module_state(pipe, 0)
        assert selfcheck["all_reduce_known_answer"], "the process group's all-reduce returned a wrong sum"

    # synthetic batches resident in HBM before the timed region
    batches = [pipe.datamanager.next_train(i) for i in range(args.steps + args.warmup)]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import neusky_amd.ops as ops
    from neusky_amd import hip as _hip
    timer = KernelTimer()
    _async_wgrad = ops.ASYNC_WGRAD
    use_graph = not args.no_graph
    skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(args.steps + args.warmup)]
    graph_note = ""
    if use_graph:
        # capture once (its eager warm-up iterations are extra and untimed); if the capture is refused, fall back to
        # host launches of the same kernels and say so in the JSON line
        try:
            stepper = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=1000)
        except Exception as exc:  # noqa: BLE001
            use_graph, graph_note = False, f" (HIP graph capture failed: {type(exc).__name__}: {str(exc)[:120]})"
            torch.cuda.synchronize()
    if use_graph:
        for i in range(args.warmup):
            stepper.step(1000 + i, batches[i][0], batches[i][1], skies[i])
        barrier()
        power = PowerSampler(dev_index)
        with power:
            t0 = time.perf_counter()
            for i in range(args.steps):
                j = args.warmup + i
                loss, _, _ = stepper.step(2000 + i, batches[j][0], batches[j][1], skies[j])
            barrier()
            dt = time.perf_counter() - t0
        # HIP events cannot be read back from inside a replayed graph: the heavy kernels' launches are timed with events on
        # one extra EAGER iteration of the same step (same kernels, shapes and stream) right after the timed region
        timer.install()
        pipe.model.second_stream = False  # one stream: every timed kernel has the chip to itself
        ops.ASYNC_WGRAD = False           # (the weight-gradient launches too: in line, not beside the next nodes' kernels)
        for it in range(3):  # (three launches per kernel: a single launch's event time varies by 10 % from run to run)
            train_iteration(pipe, opt, 3000 + it, ray_bundle=batches[-1][0], batch=batches[-1][1])
        torch.cuda.synchronize()
        pipe.model.second_stream = True
        ops.ASYNC_WGRAD = _async_wgrad
        timer.uninstall()
    else:
        for i in range(args.warmup):
            rb, b = batches[i]
            train_iteration(pipe, opt, 1000 + i, ray_bundle=rb, batch=b)
        barrier()
        power = PowerSampler(dev_index)
        with power:
            t0 = time.perf_counter()
            for i in range(args.steps):
                rb, b = batches[args.warmup + i]
                loss, _, _ = train_iteration(pipe, opt, 2000 + i, ray_bundle=rb, batch=b)
            barrier()
            dt = time.perf_counter() - t0
        timer.install()
        pipe.model.second_stream = False
        ops.ASYNC_WGRAD = False
        for it in range(3):
            train_iteration(pipe, opt, 3000 + it, ray_bundle=batches[-1][0], batch=batches[-1][1])
        torch.cuda.synchronize()
        pipe.model.second_stream = True
        ops.ASYNC_WGRAD = _async_wgrad
        timer.uninstall()
    t = torch.tensor([dt], device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss)

    # the same step with EXACT fp32 products everywhere (v_mfma_f32_32x32x2_f32, per-layer kernels): a few eager steps, rank 0
    exact = None
    if rank == 0 and world == 1 and not args.no_exact_f32 and ops._POLICY != "f32":  # (N = 1 only: a train iteration all-reduces, and the other ranks are done)
        policy = ops._POLICY
        ops.set_precision_policy("f32")
        try:
            for i in range(2):
                train_iteration(pipe, opt, 4000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(3):
                train_iteration(pipe, opt, 4002 + i, ray_bundle=batches[i][0], batch=batches[i][1])
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / 3 * 1e3
            exact = {"ms_per_step": ms, "rays_per_s": RAYS / (ms * 1e-3), "launch": "eager",
                     "note": "NSKY_PRECISION=f32: every product of the step on the exact-fp32 MFMA (157.3 TFLOP/s peak); per-layer dense kernels, no fused chains"}
        finally:
            ops.set_precision_policy(policy)

    if rank == 0:
        rays_total = RAYS * world * args.steps
        kernels = timer.summary()
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
        for k in kernels:
            k["frac_of_833_tflops"] = k["achieved_tflops"] / peak
        # a family's share of the step = its launches of ONE iteration (the timing iterations are three repeats of the same step)
        iters = 3
        for k in kernels:
            k["ms_per_step"] = k["total_ms"] / iters
            k["frac_of_8000_GBs"] = k["achieved_GBs"] / HBM_PEAK_GBS if "achieved_GBs" in k else None
        traffic, whole_step_gb, tsrc = None, None, os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
        traffic_src = {"file": "profiles/" + PMC_TRAFFIC_FILE, "collected": "committed rocprofv3 --pmc passes (tools/pmc_bench.sh), not this run"}
        dom = kernels[0]
        tj = json.load(open(tsrc)) if os.path.exists(tsrc) else None
        live_note = "not attempted (--no-live-pmc / --no-extra-configs / N > 1)"
        if world == 1 and not args.no_live_pmc and not args.no_extra_configs:
            live, live_note = live_pmc_traffic()
            if live is not None:
                committed_gb = tj.get("whole_step_GB") if tj is not None else None
                tj = dict(live, kernel_sources_sha=_sources_sha(step_kernel_sources()))
                traffic_src = {"collected": live_note, "committed_file_whole_step_GB": committed_gb, "file": "profiles/" + PMC_TRAFFIC_FILE + " (for comparison)"}
        traffic_src["live_collection"] = live_note
        if tj is not None:  # HBM bytes per launch from the rocprofv3 --pmc passes; the check that they describe THESE kernels is a field
            traffic = tj.get("bytes_per_launch", {}).get(dom["kernel"].split(" ")[0])
            whole_step_gb = tj.get("whole_step_GB")
            sha = _sources_sha(step_kernel_sources())
            traffic_src.update({"kernel_sources_sha": sha, "counters_kernel_sources_sha": tj.get("kernel_sources_sha"),
                                "describes_these_kernels": tj.get("kernel_sources_sha") == sha})
            for k in kernels:
                k["pmc_bytes_per_launch"] = tj.get("bytes_per_launch", {}).get(k["kernel"].split(" ")[0])
                m = (tj.get("mfma") or {}).get(k["kernel"].split(" ")[0]) or {}
                k["mfma_busy"], k["clock_mhz_under_counters"] = m.get("mfma_busy"), m.get("clock_mhz")

        def bound_of(k):
            hb = k.get("algorithmic_bytes_per_launch")
            if hb is not None and hb / (HBM_PEAK_GBS * 1e9) > k["algorithmic_flops_per_launch"] / (peak * 1e12):
                return "hbm"
            return "mfma"

        roof = {"bound": "mfma", "achieved": dom["achieved_tflops"], "peak": peak, "unit": "TFLOP/s", "frac": dom["achieved_tflops"] / peak}
        if bound_of(dom) == "hbm":
            roof = {"bound": "hbm", "achieved": dom["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["achieved_GBs"] / HBM_PEAK_GBS}
        top3 = [{"kernel": k["kernel"], "ms_per_step": k["ms_per_step"], "launches_per_step": k["launches"] / iters, "bound": bound_of(k),
                 "mfma_frac_of_833_tflops": k["frac_of_833_tflops"], "hbm_frac_of_8000_GBs": k["frac_of_8000_GBs"],
                 "pmc_bytes_per_launch": k.get("pmc_bytes_per_launch"),
                 "mfma_busy": k.get("mfma_busy"), "clock_mhz": k.get("clock_mhz_under_counters")} for k in kernels[:3]]
        ms_step = 1e3 * dt / args.steps
        step_roof = {"algorithmic_tflop": STEP_ALGORITHMIC_TFLOP, "tf_s": STEP_ALGORITHMIC_TFLOP / (ms_step * 1e-3),
                     "frac_of_833": STEP_ALGORITHMIC_TFLOP / (ms_step * 1e-3) / peak,
                     "GB": whole_step_gb, "tb_s": None if whole_step_gb is None else whole_step_gb / ms_step,
                     "frac_of_8": None if whole_step_gb is None else whole_step_gb / ms_step / (HBM_PEAK_GBS * 1e-3),
                     "note": "algorithmic FLOPs of one step (BASELINE.md section 4: forward 762 G x 3) and the whole step's counter traffic, both over ms_per_step"}
        fwd = "fp16 hi + residual split, 3 x v_mfma_f32_32x32x16_f16 per product on power-of-two pre-scaled operands, one fp32 accumulator (~2^-22): FiLM-SIREN chains and the SDF / albedo field alike"
        bwd = ("the same fp16 split on per-row / per-matrix pre-scaled gradients (fp32-grade) for the FiLM-SIREN chains, the SDF / albedo field and all their weight gradients; "
               "proposal layers and all N <= 64 heads: exact fp32 MFMA")
        line = {
            "metric": "train rays/sec on NeRF-OSR lk2 @1024 rays x 96 samples",
            "value": rays_total / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"f32 in memory; forward products: {fwd}; backward products: {bwd}", "data": "synthetic",
            "config": {"workload": "full NeuSky train step (BASELINE configs[2]): 1024 rays/GPU x 96 samples, proposal 256+96, "
                                   "512 illumination directions (256 upper-hemisphere DDF queries/ray), illumination decoder = this project's FiLM-SIREN H = 128 "
                                   "(RENI-shaped: latent 100x3; the reference's configured Attention decoder is absent from its tree), "
                                   "hash L16 F2 T2^19 x2, 256-wide MLPs; fwd + losses + bwd + all-reduce + 5 Adam groups",
                       "rays_per_gpu": RAYS, "samples_per_ray": SAMPLES, "illumination_directions": DIRECTIONS,
                       "parallelism": f"ray-sharded dp{world}", "final_loss": final_loss, "launcher": launcher,
                       "launch": "HIP graph replay (1 graph/step + all-reduce + 5 Adam launches)" if use_graph else "eager (host launches every kernel)" + graph_note},
            "roofline": {**roof, "traffic": traffic,
                         "traffic_unit": "bytes/launch (2*FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                         "kernel": dom["kernel"] + " = the kernel family with the largest total time per step in the three eager timing iterations "
                                                   "(every launch of every family timed; per launch: the median of the three)",
                         "top3": top3, "step": step_roof,
                         "peak_note": ("HBM3E ~8 TB/s; achieved = algorithmic bytes (inputs, saved activations and outputs once each) / launch time; "
                                       "the kernel's byte floor exceeds its flop floor at 833.3 TFLOP/s") if roof["bound"] == "hbm" else
                                      "fp16 dense MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-grade product = 833.3 TFLOP/s of algorithmic FLOPs",
                         "algorithmic_bytes_per_launch": dom.get("algorithmic_bytes_per_launch"),
                         "precision_policy": ops._POLICY, "launches_timed": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"],
                         "algorithmic_flops_per_launch": dom["algorithmic_flops_per_launch"],
                         "executed_flops_per_launch": dom["executed_flops_per_launch"]},
            "kernels": kernels[:8],
            "device_state": power.summary(),  # package power / shader clock level sampled during the timed region
            "fp32_exact": exact,
            # the process group the step's gradient exchange runs on (0 ranks: --no-spawn, no group) and its collectives exercised
            # before the timed region (rccl_selfcheck)
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 0,
            "dist_backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if dist.is_initialized() else None,
            "rccl_selfcheck": selfcheck if selfcheck is not None else pg_note,
        }
        if world == 1 and not args.no_extra_configs:
            line["envmap_decode"] = envmap_decode_line(device)
            line["attention_decoder"] = attention_decoder_line(device)
            line["global_batch_8192_one_gpu"] = global_batch_line(device)
            line["trainer_surface"] = trainer_surface_line(device)
            line["forward_only"] = forward_only_line(pipe, device)
            line["render_1080p"] = render_1080p_line(pipe, device)
            if not args.no_live_pmc:  # the saves-free forward's HBM traffic, from counters
                rp, note = render_pmc_traffic()
                line["render_1080p"]["pmc_traffic"] = rp if rp is not None else {"error": note}
                if rp is not None:
                    gb = rp["GB_per_1080p_frame_scaled_x16"]
                    line["render_1080p"]["hbm"] = {"GB_per_frame": gb, "tb_s": gb / line["render_1080p"]["ms_per_frame"],
                                                   "frac_of_8": gb / line["render_1080p"]["ms_per_frame"] / (HBM_PEAK_GBS * 1e-3)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        line_out.write(json.dumps(line) + "\n")
        line_out.flush()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
