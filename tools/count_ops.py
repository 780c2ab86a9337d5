"""count device kernels per train step by name prefix with torch.profiler (run on the GPU box)"""
import sys, os, collections
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT+"/tests", ROOT+"/tests/golden"): sys.path.insert(0,p)
import torch, time
import bench
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from util_step import randomise
pipe = bench.build_pipeline("cuda:0", 1, 0); randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches=[pipe.datamanager.next_train(i) for i in range(6)]
for i in range(3): train_iteration(pipe, opt, 1000+i, ray_bundle=batches[i][0], batch=batches[i][1])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    t0=time.perf_counter()
    train_iteration(pipe, opt, 2000, ray_bundle=batches[4][0], batch=batches[4][1])
    torch.cuda.synchronize()
    print("step wall ms", (time.perf_counter()-t0)*1e3)
ev=prof.key_averages()
rows=sorted(ev, key=lambda e:-e.self_cpu_time_total)[:45]
for e in rows: print(f"{e.key[:60]:60s} n={e.count:5d} self_cpu_ms={e.self_cpu_time_total/1e3:8.2f}")
print("total aten ops", sum(e.count for e in ev if e.key.startswith('aten::')))
