"""BASELINE config 2 (index 1): NeuSky SDF + albedo forward only, 1024 rays x 96 samples (98304 points), hash grid L=16 F=2.
Times (HIP events on the launch stream, no autograd): the whole field pass `field_values` (hash encode with tangents ->
fused geometry net + SDF normals -> colour net) and its first kernel `encode_fwd` alone, the latter against the HBM roofline:
algorithmic bytes per point = 12 (position) + L*8 corners*F*4 (gathered table entries) + output row written (value row plus
three tangent rows).  Prints ONE JSON line; run on the GPU box:  python tools/bench_forward_only.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neusky_amd import hip
from neusky_amd.utils.randomise import randomise

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
dev = "cuda:0"
rays, samples = 1024, 96
P = rays * samples
pipe = bench.build_pipeline(dev, 1, 0)
randomise(pipe)
field = pipe.model.field
g = torch.Generator().manual_seed(0)
x = ((torch.rand(P, 3, generator=g) * 2 - 1) * 0.8).to(dev)


def timed(fn, iters=20, warm=3):
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


with torch.no_grad():
    pipe.model.begin_step()
    whole = timed(lambda: field.field_values(x, want_albedo=True))
    geom, table = field.geom, field.encoding.table
    width = 3 + 6 * 6 + geom.n_levels * 2
    ldy = (width + 3) // 4 * 4
    Y = torch.empty(P, ldy, device=dev)
    T = torch.empty(3, P, ldy, device=dev)
    enc_t = timed(lambda: hip.encode_fwd(geom, table, x, field.grid_mode, True, 6, 5.0, Y, T))
    enc = timed(lambda: hip.encode_fwd(geom, table, x, field.grid_mode, True, 6, 5.0, Y, None))
gather = geom.n_levels * 8 * 2 * 4
bytes_t = P * (12 + gather + 4 * ldy * 4)
bytes_v = P * (12 + gather + ldy * 4)
print(json.dumps({
    "config": "SDF+albedo forward only, 1024 rays x 96 samples, hash L=16 F=2 (BASELINE configs[1])", "points": P,
    "field_forward_ms": whole, "points_per_s": P / whole * 1e3, "rays_per_s": rays / whole * 1e3,
    "encode_fwd_with_tangents": {"ms": enc_t, "algorithmic_bytes": bytes_t, "achieved_GBs": bytes_t / enc_t / 1e6,
                                 "peak_GBs": HBM_PEAK_GBS, "frac": bytes_t / enc_t / 1e6 / HBM_PEAK_GBS},
    "encode_fwd_values_only": {"ms": enc, "algorithmic_bytes": bytes_v, "achieved_GBs": bytes_v / enc / 1e6,
                               "peak_GBs": HBM_PEAK_GBS, "frac": bytes_v / enc / 1e6 / HBM_PEAK_GBS},
    "note": "98304 points is ~384 workgroups of 256: one and a half waves of the chip; the kernel is launch/latency bound at this size, "
            "the table (L2/MALL resident after the first touch) is gathered from cache, so the HBM fraction is a lower bound on nothing "
            "but the size of the problem"}))
