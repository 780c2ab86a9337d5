"""Per-shape breakdown of every GEMM launch in one eager train step (run on the GPU box).

Groups launches by (M, N, K, a_kcontig, b_kcontig, epilogue, precision, k_splits) and prints count, total ms,
achieved TFLOP/s and the algorithmic HBM bytes of A/B/C/aux so memory-bound shapes stand out."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import torch
import bench
from neusky_amd import hip
import neusky_amd.ops as ops
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from util_step import randomise

pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(6)]
for i in range(3):
    train_iteration(pipe, opt, 1000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
torch.cuda.synchronize()

records = []
orig = hip.gemm


def timed(A, B, Cout, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(A, B, Cout, M, N, K, **kw)
    e1.record()
    naux = sum(1 for k in ("aux0", "aux1", "aux2", "out1", "out2") if kw.get(k) is not None)
    key = (M, N, K, int(bool(kw.get("a_kcontig", True))), int(bool(kw.get("b_kcontig", True))), int(kw.get("epi", 0)),
           int(kw.get("precision", 0)), int(kw.get("k_splits", 1)), naux, int(kw.get("row_mod", 0) or 0))
    records.append((key, e0, e1))
    return out


orig_planes = hip.gemm_planes


def timed_planes(A, planes, Cout, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig_planes(A, planes, Cout, M, N, K, **kw)
    e1.record()
    naux = sum(1 for k in ("aux0", "aux1", "aux2", "out1", "out2") if kw.get(k) is not None)
    key = (M, N, K, 1, 9, int(kw.get("epi", 0)), int(kw.get("precision", 0)), 1, naux, int(kw.get("row_mod", 0) or 0))  # bkc 9 = planes
    records.append((key, e0, e1))
    return out


hip.gemm = timed
hip.gemm_planes = timed_planes
e_all0, e_all1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e_all0.record()
train_iteration(pipe, opt, 2000, ray_bundle=batches[4][0], batch=batches[4][1])
e_all1.record()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, a, b in records:
    n, ms = agg.get(key, (0, 0.0))
    agg[key] = (n + 1, ms + a.elapsed_time(b))
tot = sum(ms for _, ms in agg.values())
print(f"eager step {e_all0.elapsed_time(e_all1):.2f} ms, {len(records)} gemm launches, {tot:.2f} ms in gemm (event-bracketed)")
print("     M     N     K akc bkc epi prec ks naux rmod |   n   tot_ms  avg_us  TFLOP/s  min_GB  GB/s")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, akc, bkc, epi, prec, ks, naux, rmod = key
    fl = 2.0 * M * N * K * n
    byts = 4.0 * (M * K + N * K + M * N * (1 + naux)) * n
    print(f"{M:7d} {N:5d} {K:5d} {akc:3d} {bkc:3d} {epi:3d} {prec:4d} {ks:2d} {naux:4d} {rmod:5d} | {n:3d} {ms:8.3f} {ms / n * 1e3:7.1f} "
          f"{fl / ms / 1e9:8.1f} {byts / n / 1e9:7.3f} {byts / ms / 1e6:6.0f}")
