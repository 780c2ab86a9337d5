"""bench.py's launch forms on the one GPU a test box has.

* `python bench.py --gpus N` WITHOUT an external launcher: the parent starts one child per rank before any GPU call (spawn_ranks), relays
  rank 0's single JSON line and fails if any rank does.  N = 1 runs over RCCL ("nccl", one-rank communicator: the broadcast, a known-answer
  all-reduce and the gradient slab's all-reduce really execute); N = 2 shares cuda:0 over gloo (NSKY_BENCH_DEVICE=0, NSKY_DIST_BACKEND=gloo).
* the driver's contract for N > 1, `python -m torch.distributed.run ... bench.py --gpus N`: same file, ranks started by torchrun.

Both check that nothing after the timed region needs a collective the other rank has left (the exact-fp32 / forward-only / render /
CPU-baseline legs are N = 1 only) and that the line is well formed: whole-job rays over the max-over-ranks time."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--no-cpu-baseline", "--no-exact-f32", "--no-extra-configs"]


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _check_two_ranks(d):
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "ray-sharded dp2"
    assert d["value"] > 0 and abs(d["value"] - 2 * 1024 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 1e-6 * d["value"]  # whole-job rays / max-over-ranks time
    assert d["fp32_exact"] is None and "forward_only" not in d and "cpu_baseline" not in d  # N = 1 legs
    assert d["roofline"]["frac"] > 0 and d["config"]["final_loss"] == d["config"]["final_loss"]
    assert d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo"
    assert d["rccl_selfcheck"]["all_reduce_known_answer"] and d["rccl_selfcheck"]["tensors_broadcast"] > 50


def test_bench_spawns_its_own_two_ranks_on_one_gpu():
    env = dict(os.environ, NSKY_BENCH_DEVICE="0", NSKY_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(out)
    _check_two_ranks(d)
    assert d["config"]["launcher"] == "self-spawned"


def test_bench_one_rank_through_the_spawn_path_over_rccl():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "NSKY_BENCH_DEVICE", "NSKY_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"] + FAST,
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(out)
    assert d["n_gpus"] == 1 and d["config"]["launcher"] == "self-spawned" and d["config"]["parallelism"] == "ray-sharded dp1"
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl (RCCL)", d.get("rccl_selfcheck")
    sc = d["rccl_selfcheck"]
    assert sc["backend"] == "nccl" and sc["all_reduce_known_answer"] and sc["slab_stays_zero"] and sc["tensors_broadcast"] > 50
    assert sc["gradient_slab_bytes"] > 100e6 and sc["gradient_slab_all_reduce_ms"] > 0  # the ~110 MB slab of all five groups
    assert abs(d["value"] - 1024 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # a rank that fails takes the job down with a non-zero exit code (no hang, no JSON line)
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"] + FAST,
                         cwd=ROOT, env=dict(env, NSKY_BENCH_DEVICE="99"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]


def test_bench_two_ranks_under_torchrun_one_gpu():
    env = dict(os.environ, NSKY_BENCH_DEVICE="0", NSKY_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(out)
    _check_two_ranks(d)
    assert d["config"]["launcher"] == "external (torch.distributed.run)"
