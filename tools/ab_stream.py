import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.getcwd() + "/tests"); sys.path.insert(0, os.getcwd() + "/tests/golden")
import torch, bench
from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
from neusky_amd.utils.randomise import randomise
torch.manual_seed(1234)
pipe = bench.build_pipeline("cuda:0", 1, 0)
randomise(pipe, seed=0)
opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
batches = [pipe.datamanager.next_train(i) for i in range(40)]
skies = [pipe.datamanager.get_sky_ray_bundle(pipe.config.num_sky_rays) for _ in range(40)]
for second in (True, False, True, False):
    pipe.model.second_stream = second
    st = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=1000)
    for i in range(5): st.step(1000 + i, batches[i][0], batches[i][1], skies[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(30): st.step(2000 + i, batches[5 + i][0], batches[5 + i][1], skies[5 + i])
    torch.cuda.synchronize()
    print("second_stream", second, round((time.perf_counter() - t0) / 30 * 1e3, 3), "ms/step", flush=True)
