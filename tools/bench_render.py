"""Relighting render pass (BASELINE config 5): one 1920x1080 frame, 512 illumination directions (256 upper-hemisphere DDF
visibility queries per ray), static chunks replayed from a HIP graph.  Prints ms/frame and rays/s."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"): sys.path.insert(0, p)
import torch
import bench
from util_step import randomise
dev = "cuda:0"
H, W = (1080, 1920) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
chunk = int(os.environ.get("NSKY_RENDER_CHUNK", "4096"))
pipe = bench.build_pipeline(dev, 1, 0)
randomise(pipe)
pipe.eval()
from neusky_amd.cameras.rays import RayBundle
g = torch.Generator().manual_seed(0)
ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
d_cam = torch.stack([(xs - W / 2) / 1100.0, (ys - H / 2) / 1100.0, torch.ones(H, W)], -1)
cR, cp = pipe.datamanager.cam_R[0], pipe.datamanager.cam_pos[0]
d = torch.einsum("ij,hwj->hwi", cR, d_cam)
d = d / d.norm(dim=-1, keepdim=True)
rb = RayBundle(origins=cp.expand(H, W, 3).contiguous().to(dev), directions=d.to(dev), pixel_area=torch.ones(H, W, 1, device=dev),
               camera_indices=torch.zeros(H, W, 1, dtype=torch.long, device=dev), metadata={"directions_norm": torch.ones(H, W, 1, device=dev)})
modes = {"graph": (True,), "eager": (False,)}.get(sys.argv[3] if len(sys.argv) > 3 else "", (True, False))  # (eager: one dispatch record per launch for --pmc passes)
for use_graph in modes:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe.model.get_outputs_for_camera_ray_bundle(rb, camera_index=0, chunk=chunk, use_graph=use_graph)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"frame": f"{W}x{H}", "chunk_rays": chunk, "hip_graph": use_graph, "ms_per_frame": dt * 1e3, "rays_per_s": H * W / dt,
                      "rgb_mean": float(out["rgb"].mean())}))
