"""A few short training runs from different seeds (graph-replayed steps on the bench workload): the loss must stay finite and fall
from every start (run on the GPU box): python tools/seed_sweep.py [seeds] [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
from neusky_amd.utils.randomise import randomise
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
bad = 0
for seed in range(1, n_seeds + 1):
    torch.manual_seed(seed)
    pipe = bench.build_pipeline("cuda:0", 1, 0)
    randomise(pipe, seed=seed)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    b = [pipe.datamanager.next_train(i) for i in range(4)]
    sk = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(4)]
    st = GraphedTrainStep(pipe, opt, b[0][0], b[0][1], warmup=2, start_step=0)
    tr = []
    for i in range(steps):
        loss, _, _ = st.step(3 + i, b[i % 4][0], b[i % 4][1], sk[i % 4])
        tr.append(float(loss))
    ok = all(v == v and abs(v) < 1e6 for v in tr) and tr[-1] < tr[0]
    bad += not ok
    print("seed", seed, "ok" if ok else "BAD", [round(v, 3) for v in tr[:3]], "...", [round(v, 3) for v in tr[-3:]], flush=True)
    del st, opt, pipe
    torch.cuda.empty_cache()
assert bad == 0, f"{bad} runs diverged"
print("all ok")
