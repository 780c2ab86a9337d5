"""Longer run of the graph-replayed train step on the synthetic workload: loss must stay finite and fall (run on the GPU box).
    python tools/train_sanity.py [steps]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import torch, bench
from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
from util_step import randomise
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
torch.manual_seed(0)
pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
nb = 16
batches = [pipe.datamanager.next_train(i) for i in range(nb)]
skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(nb)]
stepper = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=0)
hist = []
t0 = time.perf_counter()
for i in range(steps):
    j = i % nb
    loss, ld, _ = stepper.step(3 + i, batches[j][0], batches[j][1], skies[j])
    if i % 25 == 0 or i == steps - 1:
        v = float(loss)
        hist.append((i, v))
        print(i, round(v, 4), {k: round(float(x), 5) for k, x in ld.items() if k in ("rgb_l1_loss", "eikonal_loss", "depth_l1_loss", "sdf_level_set_visibility_loss")}, flush=True)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / steps * 1e3, " peak GB", torch.cuda.max_memory_allocated() / 1e9, " now GB", torch.cuda.memory_allocated() / 1e9)
assert all(v == v and abs(v) < 1e6 for _, v in hist), "loss diverged"
assert hist[-1][1] < 0.5 * hist[0][1], "loss did not fall"
bad = [n for n, p in pipe.named_parameters() if not torch.isfinite(p).all()]
assert not bad, bad
print("ok")
