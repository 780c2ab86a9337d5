"""worst relative gradient error per parameter tensor of the HIP step vs the float64 oracle, next to the float32 oracle's
own distance from float64 (run on the GPU box):  python tools/grad_err.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import torch
import test_gpu_step as T
step = T.compute_step()
got = T._module_grads(step["pipe"])
rows = []
for k, ref in step["grads"].items():
    if ref is None or k.startswith("reni."):
        continue
    b = ref.reshape(-1)
    sc = b.abs().max().item() + 1e-30
    e_hip = (got[k].detach().cpu().double().reshape(-1) - b).abs().max().item() / sc
    e_f32 = (step["grads32"][k].double().reshape(-1) - b).abs().max().item() / sc
    rows.append((e_hip, e_f32, k))
for e_hip, e_f32, k in sorted(rows, reverse=True)[:14]:
    print(f"{k:24s} hip {e_hip:.2e}   fp32-oracle {e_f32:.2e}")
