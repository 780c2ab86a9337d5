import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import torch
import bench
from util_step import randomise
from neusky_amd.cameras.rays import RayBundle
dev = "cuda:0"
pipe = bench.build_pipeline(dev, 1, 0); randomise(pipe); pipe.eval()
H, W = 128, 256
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
d_cam = torch.stack([(xs - W / 2) / 200.0, (ys - H / 2) / 200.0, torch.ones(H, W)], -1)
cR, cp = pipe.datamanager.cam_R[0], pipe.datamanager.cam_pos[0]
d = torch.einsum("ij,hwj->hwi", cR, d_cam); d = d / d.norm(dim=-1, keepdim=True)
rb = RayBundle(origins=cp.expand(H, W, 3).contiguous().to(dev), directions=d.to(dev), pixel_area=torch.ones(H, W, 1, device=dev),
               camera_indices=torch.zeros(H, W, 1, dtype=torch.long, device=dev), metadata={"directions_norm": torch.ones(H, W, 1, device=dev)})
outs = {}
for chunk in (2048, 4096, 8192, 16384):
    for use_graph in (False, True):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=0, chunk=chunk, use_graph=use_graph)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        outs[(chunk, use_graph)] = o
        print(chunk, "graph" if use_graph else "eager", f"{dt*1e3:.1f} ms (2nd run)", "rgb mean", float(o["rgb"].mean()), "peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
    torch.cuda.empty_cache()
ref = outs[(2048, False)]
for key, o in outs.items():
    print(key, {k: float((o[k] - ref[k]).abs().max()) for k in ("rgb", "p2p_dist", "normal", "accumulation")})
