"""per-term losses of the first eager train steps (seeded): a quick A/B of arithmetic-neutral changes (run on the GPU box)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tests/golden")
import torch, bench
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
from util_step import randomise
pipe = bench.build_pipeline("cuda:0", 1, 0); torch.manual_seed(1234); randomise(pipe, seed=0)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(4)]
torch.manual_seed(7)
for i in range(4):
    loss, ld, _ = train_iteration(pipe, opt, 1000 + i, ray_bundle=batches[i][0], batch=batches[i][1])
    print(i, float(loss), {k: round(float(v), 6) for k, v in ld.items()})
