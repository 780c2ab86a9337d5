"""ops.retire_graph (the workaround of the HIP runtime's graph-destroy use-after-free) and ops.role_stream (one stream
per role: pool streams alias after 32 creations): host logic on the CPU, the real thing on the GPU."""
import gc
import time

import pytest
import torch


class _Graph:
    alive = 0

    def __init__(self):
        _Graph.alive += 1

    def __del__(self):
        _Graph.alive -= 1


def test_retire_keeps_a_graph_alive_and_destroys_it_later(monkeypatch):
    from neusky_amd import ops
    calls = []
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: calls.append("sync"))
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(ops, "RETIRE_SECONDS", 0.5)
    ops._RETIRED_GRAPHS.clear()
    _Graph.alive = 0
    ops.retire_graph(_Graph())
    gc.collect()
    assert _Graph.alive == 1 and calls == [], "a retired graph must outlive the call that retires it (its last launch may still be in flight)"
    ops.retire_graph(None)  # nothing is old enough yet
    assert _Graph.alive == 1
    time.sleep(0.6)
    ops.retire_graph(_Graph())  # a later retirement: the old one is destroyed HERE, after a device synchronize, on this thread
    gc.collect()
    assert _Graph.alive == 1 and calls == ["sync"] and len(ops._RETIRED_GRAPHS) == 1
    # never from inside a capture (a synchronize would invalidate it): the purge waits for the next call
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: True)
    time.sleep(0.6)
    ops.retire_graph(None)
    assert _Graph.alive == 1 and calls == ["sync"]
    ops._RETIRED_GRAPHS.clear()


@pytest.mark.gpu
def test_role_streams_are_distinct_and_survive_pool_wraparound():
    from neusky_amd import ops
    roles = {r: ops.role_stream(r) for r in ops.ROLES}
    assert len({s.stream_id for s in roles.values()}) == len(ops.ROLES)
    others = [torch.cuda.Stream() for _ in range(70)]  # the pool (32 per priority) wraps twice: some of these ARE the role streams
    assert any(o == roles["capture"] for o in others), "torch.cuda.Stream() objects alias pool streams: the reason the roles are created once"
    again = {r: ops.role_stream(r) for r in ops.ROLES}
    assert all(again[r] is roles[r] for r in ops.ROLES)


@pytest.mark.gpu
def test_captures_run_on_the_capture_role_stream_after_many_streams():
    """the failing case of the round in small: a process that has created many streams builds a pipeline and captures its step"""
    from util_step import randomise, small_pipeline_config
    from neusky_amd import ops
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    _ = [torch.cuda.Stream() for _ in range(45)]
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=32, S=8, D=24, images=4).setup(device="cuda:0")
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    rb, batch = pipe.datamanager.next_train(0)
    stepper = GraphedTrainStep(pipe, opt, rb, batch, warmup=1, start_step=10)
    m = pipe.model
    ids = {ops.role_stream("capture").stream_id, m._illumination_stream().stream_id, m._ddf_fit_stream().stream_id, ops.role_stream("wgrad").stream_id}
    assert len(ids) == 4
    loss = stepper.step(11, rb, batch)[0]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(loss))
