"""stress of the path tests/test_gpu_eval_methods.py walks (fresh small pipeline, eval-latent fit under a captured graph, eval forward, frame render),
N times in one process: python tools/flake_eval.py [N]   (hunting an intermittent abort; NSKY_LIB / NSKY_FILM_BWD / NSKY_ASYNC_WGRAD select variants)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
from util_step import randomise, small_pipeline_config
from neusky_amd.engine import Optimizers, neusky_optimizers, train_iteration
DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(n):
    torch.manual_seed(it)
    if it % 2 == 0:  # a training step first, like the tests that run before (side-stream weight gradients, slabs)
        pipe = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, images=4).setup(device=DEV)
        pipe.train(); randomise(pipe)
        opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
        rb, b = pipe.datamanager.next_train(0)
        for s in range(2):
            train_iteration(pipe, opt, 100 + s, ray_bundle=rb, batch=b)
        del pipe, opt
    cfg = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, images=4)
    cfg.model.eval_latent_optimizer = {"lr": 1e-1, "eps": 1e-15, "lr_final": 1e-7, "max_steps": 4}
    cfg.datamanager.eval_num_rays_per_batch = 64
    cfg.datamanager.eval_image_height, cfg.datamanager.eval_image_width = 12, 16
    pipe = cfg.setup(device=DEV)
    pipe.train(); randomise(pipe)
    outs, loss_dict, metrics = pipe.get_eval_loss_dict(step=7)
    m, images = pipe.get_eval_image_metrics_and_images(step=7)
    torch.cuda.synchronize()
    print(it, float(sum(loss_dict.values())), flush=True)
    del pipe
print("ok")
