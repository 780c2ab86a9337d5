"""The HIP-graph-captured training iteration: replays run, are finite, move every parameter group, and agree with the
eager path on a fixed input when the in-graph random draws are irrelevant (statistically: same loss scale)."""
import pytest
import torch

from util_step import randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_graphed_train_step_replays():
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers, train_iteration
    torch.manual_seed(0)
    pipe = small_pipeline_config(R=64, num_prop=(32, 16), S=12, D=32, vmf=(2, 16), sky=16, images=7).setup(device=DEV)
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    batches = [pipe.datamanager.next_train(i) for i in range(8)]
    eager_loss, _, _ = train_iteration(pipe, opt, 100, ray_bundle=batches[0][0], batch=batches[0][1])
    before = {g.name: g.flat_p.clone() for g in opt.groups}
    stepper = GraphedTrainStep(pipe, opt, batches[1][0], batches[1][1], warmup=2, start_step=101)
    losses = []
    for i in range(5):
        rb, b = batches[2 + i]
        loss, loss_dict, _ = stepper.step(110 + i, rb, b)
        losses.append(float(loss))
        assert all(torch.isfinite(v) for v in loss_dict.values())
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert 0.2 < losses[0] / float(eager_loss) < 5.0, (losses, float(eager_loss))
    for g in opt.groups:  # every group received gradients through the graph and was stepped
        assert (g.flat_p - before[g.name]).abs().max().item() > 0, g.name
    # different inputs give different losses: the static input buffers really feed the graph
    assert len({round(l, 6) for l in losses}) > 1
