"""python tools/bench_step.py [steps] : ms per graph-replayed train step at the bench's size, nothing else (same-box A/B runs of the lab
switches of tools/lab.py: NSKY_ASYNC_WGRAD=0, NSKY_FIT_STREAM=0, NSKY_FILM_ASYNC=0)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
import bench
from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
from neusky_amd.utils.randomise import randomise

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = "cuda:0"
torch.cuda.set_device(0)
torch.manual_seed(1234)
pipe = bench.build_pipeline(dev, 1, 0)
randomise(pipe, seed=0)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(4)]
skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(4)]
stepper = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=1000)
for i in range(5):
    stepper.step(1000 + i, batches[i % 4][0], batches[i % 4][1], skies[i % 4])
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(steps):
        loss, _, _ = stepper.step(2000 + i, batches[i % 4][0], batches[i % 4][1], skies[i % 4])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / steps * 1e3)
print(json.dumps({"ms_per_step": min(ts), "all": ts, "loss": float(loss),
                  "switches": {k: v for k, v in os.environ.items() if k.startswith("NSKY_")}}))
