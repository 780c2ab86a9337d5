"""bench.py's own N > 1 path (the launch contract of the driver's scaling run: torch.distributed.run, one rank per process, barrier +
max-over-ranks timing, rank 0 prints ONE JSON line whose value is the whole job's rays/s) on the one GPU a test box has: two ranks share
cuda:0 over gloo (NSKY_BENCH_DEVICE=0, NSKY_DIST_BACKEND=gloo).  Checks that nothing after the timed region needs a collective the other
rank has left (the exact-fp32 / forward-only / render / CPU-baseline legs are N = 1 only) and that the line is well formed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, NSKY_BENCH_DEVICE="0", NSKY_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "ray-sharded dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1024 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-6 * d["value"]  # whole-job rays / max-over-ranks time
    assert d["fp32_exact"] is None and "forward_only" not in d and "cpu_baseline" not in d  # N = 1 legs
    assert d["roofline"]["frac"] > 0 and d["config"]["final_loss"] == d["config"]["final_loss"]
