"""which piece of the evaluation path aborts now and then?  python tools/flake_eval2.py <fit|fit_eager|render|render_eager|forward> [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p_)
import torch
from util_step import randomise, small_pipeline_config
DEV = "cuda:0"
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for it in range(n):
    torch.manual_seed(it)
    cfg = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, images=4)
    cfg.model.eval_latent_optimizer = {"lr": 1e-1, "eps": 1e-15, "lr_final": 1e-7, "max_steps": 4}
    cfg.datamanager.eval_num_rays_per_batch = 64
    cfg.datamanager.eval_image_height, cfg.datamanager.eval_image_width = 12, 16
    pipe = cfg.setup(device=DEV)
    pipe.train(); randomise(pipe)
    m = pipe.model
    if mode in ("fit", "fit_eager"):
        m.fit_latent_codes_for_eval(pipe.datamanager, global_step=7, use_graph=(mode == "fit"))
    elif mode in ("render", "render_eager"):
        pipe.eval()
        idx, cam_rb, full = pipe.datamanager.next_eval_image(0)
        m.get_outputs_for_camera_ray_bundle(cam_rb, camera_index=0, use_graph=(mode == "render"))
    else:
        pipe.eval()
        rb, batch = pipe.datamanager.next_eval(0)
        m(rb)
    torch.cuda.synchronize()
    del pipe, m
    if it % 10 == 9:
        print(it, flush=True)
print("ok")
