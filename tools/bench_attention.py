"""the train step with the reference's configured RENI++ attention decoder (bench.py's `attention_decoder` key) on its own: for
rocprofv3 --kernel-trace --stats -- python3 tools/bench_attention.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, ROOT + "/tests", ROOT + "/tests/golden"):
    sys.path.insert(0, p)
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__))); import lab; lab.apply()  # NSKY_* lab switches (tools/lab.py)
import torch
import bench

torch.cuda.set_device(0)
print(json.dumps(bench.attention_decoder_line("cuda:0", steps=int(sys.argv[1]) if len(sys.argv) > 1 else 10)))
