"""One long-lived process doing what one `ns-train neusky` run does (neusky_pipeline.py:204-210 -> neusky_model.py:1503-1588): train
replays, eval-latent fits with their own captured graph, frame renders on chunk graphs, more training, the pipeline dropped, a second,
differently sized pipeline doing the same.  This is the sequence behind the round-4/5 aborts ("free(): invalid pointer" / SIGSEGV: a
use-after-free in the HIP runtime when a captured graph is destroyed with a launch's completion callback pending -- ops.retire_graph);
with every graph retired instead of destroyed next to its last launch it runs clean.  tools/flake.sh repeats it across processes."""
import gc

import pytest
import torch

from util_step import randomise, small_pipeline_config

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _life_of_a_pipeline(R, S, D, images, conditioning):
    from neusky_amd import ops
    from neusky_amd.engine import GraphedTrainStep, Optimizers, neusky_optimizers
    torch.manual_seed(R)
    cfg = small_pipeline_config(R=R, num_prop=(2 * S, S), S=S, D=D, images=images)
    cfg.model.illumination_field.conditioning = conditioning
    cfg.model.eval_latent_optimizer = {"lr": 1e-1, "eps": 1e-15, "lr_final": 1e-7, "max_steps": 6}
    cfg.datamanager.eval_num_rays_per_batch = R
    cfg.datamanager.eval_image_height, cfg.datamanager.eval_image_width = 8, 12
    pipe = cfg.setup(device=DEV)
    pipe.train()
    randomise(pipe)
    opt = Optimizers(neusky_optimizers(), pipe.get_param_groups())
    batches = [pipe.datamanager.next_train(i) for i in range(3)]
    stepper = GraphedTrainStep(pipe, opt, batches[0][0], batches[0][1], warmup=2, start_step=100)
    losses = []
    for i in range(20):
        losses.append(stepper.step(100 + i, batches[i % 3][0], batches[i % 3][1])[0])
    for e in range(3):  # three evaluations: each fits the eval latents under a graph of its own and renders a frame on chunk graphs
        _, loss_dict, _ = pipe.get_eval_loss_dict(step=1000 + e)
        assert all(torch.isfinite(v) for v in loss_dict.values())
        m, images_ = pipe.get_eval_image_metrics_and_images(step=1000 + e)
        assert m["psnr"] == m["psnr"] and images_["img"].shape[0] == 8
        assert pipe.model.training and not pipe.model.fitting_eval_latents
    for i in range(20):
        losses.append(stepper.step(200 + i, batches[i % 3][0], batches[i % 3][1])[0])
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(l)) for l in losses)
    retired = len(ops._RETIRED_GRAPHS)
    del stepper, opt, pipe
    gc.collect()
    return retired


def test_train_fit_render_train_drop_twice_in_one_process():
    from neusky_amd import ops
    a = _life_of_a_pipeline(R=32, S=8, D=24, images=4, conditioning="FiLM")
    b = _life_of_a_pipeline(R=48, S=12, D=32, images=5, conditioning="FiLM")
    c = _life_of_a_pipeline(R=32, S=8, D=16, images=3, conditioning="Attention")
    torch.cuda.synchronize()
    x = torch.unique(torch.randint(0, 7, (4096,), device=DEV))  # the first host allocation storm behind the fits used to be where it died
    assert x.numel() == 7
    # the graphs of the fits were RETIRED (kept alive past their last replay), and old ones are destroyed by later retirements
    assert a >= 1 and b >= 1 and c >= 1 and len(ops._RETIRED_GRAPHS) < 40
