"""Every dense-kernel launch of one eager train step with its shape, precision, flags and HIP-event time (run on the GPU box)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neusky_amd import hip
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from neusky_amd.utils.randomise import randomise
pipe = bench.build_pipeline("cuda:0", 1, 0); randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(4)]
for i in range(2):
    train_iteration(pipe, opt, 1000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
torch.cuda.synchronize()
recs = []
def wrap(name):
    orig = getattr(hip, name)
    def f(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = orig(*a, **kw); e1.record()
        if name == "gemm":
            M, N, K = a[3], a[4], a[5]
            desc = f"M={M} N={N} K={K} prec={kw.get('precision', 0)} akc={int(kw.get('a_kcontig', True))} bkc={int(kw.get('b_kcontig', True))} epi={kw.get('epi', 0)} splits={kw.get('k_splits', 0)} ant={kw.get('a_native_nt', 0)} bnt={kw.get('b_native_nt', 0)}"
        elif name == "gemm_planes":
            M, N, K = a[3], a[4], a[5]
            desc = f"M={M} N={N} K={K} prec={kw.get('precision', 0)} epi={kw.get('epi', 0)}"
        else:
            desc = ""
        recs.append((name, desc, e0, e1))
        return out
    setattr(hip, name, f)
for n in ("gemm", "gemm_planes", "film_chain_fwd", "film_chain_bwd_film", "film_chain_bwd_map", "wgrad_native_batch", "encode_fwd", "encode_bwd"):
    wrap(n)
pipe.model.second_stream = False
train_iteration(pipe, opt, 2000, ray_bundle=batches[3][0], batch=batches[3][1])
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, desc, e0, e1 in recs:
    k = (name, desc)
    ms = e0.elapsed_time(e1)
    c, t = agg.get(k, (0, 0.0))
    agg[k] = (c + 1, t + ms)
tot = 0.0
for (name, desc), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print(f"{t:8.3f} ms  x{c:2d}  {name:22s} {desc}")
print("total", tot)
