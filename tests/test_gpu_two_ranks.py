"""Two ranks of the REAL pipeline on one GPU (SURVEY 8(e) correctness definition; VERDICT r2 "next round" 5): two fresh child
processes on cuda:0 with backend `gloo`, each running the train iteration on its half of a 32-ray batch with injected random
draws; the all-reduced gradient slab must equal the single-process gradient of the concatenated batch
(neusky/pipelines/neusky_pipeline.py:198-200 is the reference's DDP wrap, broken as written: SURVEY F6), and a graph-replayed
step + all-reduce + Adam must work with world_size = 2.  The children are separate programs started with subprocess (never an
exec of this process)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(world, tmp, shard=False):
    port = _free_port()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if shard:
        env["NSKY_TEST_SHARD_ILLUMINATION"] = "1"
    outs = [os.path.join(tmp, f"w{world}_r{r}{'_shard' if shard else ''}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), str(world), str(port), outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(f"--- rank {i} (exit {p.returncode})\n{l[-3000:]}" for i, (p, l) in enumerate(zip(procs, logs)))
    return np.load(outs[0])


def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path):
    two = _launch(2, str(tmp_path))
    one = _launch(1, str(tmp_path))
    keys = sorted(k for k in one.files if k.startswith("g:"))
    assert keys == sorted(k for k in two.files if k.startswith("g:")) and len(keys) > 40
    rows, bad = [], []
    for k in keys:
        a, b = two[k].astype(np.float64), one[k].astype(np.float64)
        scale = np.abs(b).max()
        err = np.abs(a - b).max()
        rows.append((k[2:], err / (scale + 1e-30), scale))
        # fp32 reduction-order noise only: the two halves' sums are added in another order (and the hash-table atomics in any order)
        if err > 2e-3 * scale + 1e-9:
            bad.append((k, err, scale))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_two_rank_grad_errors.txt", "w") as f:
        f.write("parameter  max|g(2 ranks, all-reduced) - g(1 rank, 32 rays)| / max|g|   max|g|\n")
        for k, e, s in rows:
            f.write(f"{k:64s} {e:.3e} {s:.3e}\n")
    assert not bad, bad
    # batch-mean losses: the single-process loss is the mean of the two ranks' losses only up to the terms both ranks share;
    # what must agree exactly in structure is the gradient above.  The graph-replayed step reproduces the eager slab and steps Adam.
    for w in (one, two):
        se, sg = w["slab_eager"].astype(np.float64), w["slab_graph"].astype(np.float64)
        assert np.isfinite(sg).all() and np.abs(se - sg).max() <= 2e-3 * np.abs(se).max(), np.abs(se - sg).max()
        assert int(w["moved"]) > 40, "the Adam step behind the graph replay moved too few parameters"
        assert abs(float(w["loss"]) - float(w["graph_loss"])) < 1e-4 * abs(float(w["loss"]))


def test_camera_sharded_illumination_decode_equals_one_rank(tmp_path):
    """VERDICT r5 item 5 (neusky_model.py:445-551): with `shard_illumination_decode` rank r decodes the environment light of cameras r::2
    only, the [U, D, 3] colours are all-gathered and their gradient reduce-scattered (distributed.CameraAllGather); the all-reduced
    gradient of every parameter -- the illumination latents and scales above all -- equals the one-rank gradient of the concatenated
    batch within the same bar as the unsharded two-rank run.  (gloo, both ranks on cuda:0; unmeasured on hardware.)"""
    two = _launch(2, str(tmp_path), shard=True)
    one = _launch(1, str(tmp_path))
    keys = sorted(k for k in one.files if k.startswith("g:"))
    assert keys == sorted(k for k in two.files if k.startswith("g:")) and len(keys) > 40
    bad = []
    for k in keys:
        a, b = two[k].astype(np.float64), one[k].astype(np.float64)
        scale = np.abs(b).max()
        if np.abs(a - b).max() > 2e-3 * scale + 1e-9:
            bad.append((k, float(np.abs(a - b).max()), float(scale)))
    assert not bad, bad
    lat = [k for k in keys if "illumination_latents" in k or k.endswith("train_scale")]
    assert lat and all(np.abs(one[k]).max() > 0 for k in lat), "the probe batch must reach the latents"


def test_sharded_decode_collectives_are_captured_under_rccl():
    """with `graph_replay` the sharded decode's all-gather and reduce-scatter are part of the captured step: RCCL collectives inside a HIP
    graph.  One rank is all one GPU gives, and it answers the question that could not be answered on paper -- the capture and its replays
    work on this stack (RCCL 2.26, HIP 7.0): nerfstudio's loop with graph replay, SlabAdam and the shard forced on over a one-rank
    communicator trains (tests/shard_capture_worker.py)."""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(HERE, "shard_capture_worker.py"), str(_free_port())], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "captured: True" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-2500:])
    losses = eval(out.stdout.split("losses", 1)[1].strip().splitlines()[0])
    assert len(losses) == 4 and all(l == l and l < 1e4 for l in losses) and losses[-1] < losses[0]
