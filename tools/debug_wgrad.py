import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/tests")
import torch
from neusky_amd import hip
from test_gpu_wgrad_native import _case
DEV = "cuda:0"
for (M, n_out, k_in, spread, mode) in [(4096, 256, 256, False, ""), (5000, 256, 256, False, ""), (4096, 256, 256, True, ""),
                                       (4096, 256, 256, True, "nox"), (4096, 256, 256, True, "nodz"), (8192, 256, 256, False, ""), (65536, 256, 256, False, "")]:
    dz, x = _case(M, n_out, k_in, 5, spread)
    if mode == "nox":
        x = torch.sin(torch.randn(M, k_in) * 3.0)
    if mode == "nodz":
        dz = torch.randn(M, n_out) * 1e-3
    dzd, xd = dz.to(DEV), x.to(DEV)
    A = hip.film_rows_to_native(dzd, n_out); B = hip.film_rows_to_native(xd, k_in)
    gmax = dzd.abs().max().reshape(1)
    dW = torch.zeros(n_out, k_in, device=DEV); db = torch.zeros(n_out, device=DEV)
    hip.wgrad_native(A, n_out // 32, B, k_in // 32, M, dW, db, gmax)
    ref = dzd.double().T @ xd.double(); bar = dzd.double().abs().T @ xd.double().abs()
    e = ((dW.double() - ref).abs() / bar)
    print(M, spread, mode, "worst", float(e.max()), "by k quarter", [float(e[:, i * k_in // 4:(i + 1) * k_in // 4].max()) for i in range(4)],
          "db", float(((db.double() - dzd.double().sum(0)).abs() / dzd.double().abs().sum(0)).max()))
    if spread:
        fs = dzd.abs().max(0).values
        idx = e.max(1).values.argsort(descending=True)[:5]
        print("   worst features", idx.tolist(), "their max|dz|/gmax", (fs[idx] / gmax).tolist())
