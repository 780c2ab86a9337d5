"""Worker of tests/test_gpu_two_ranks.py::test_sharded_decode_collectives_are_captured_under_rccl: can the camera-sharded illumination decode's
all-gather / reduce-scatter be CAPTURED with RCCL (they are part of the replayed step under graph_replay)?  One rank, one GPU: a one-rank process
group ("nccl"), the shard forced on with (rank 0, world 1), nerfstudio's loop with the pipeline's graph replay.
    python tests/shard_capture_worker.py <port>"""
import datetime, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE, os.path.join(HERE, "golden")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1] if len(sys.argv) > 1 else "29731")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
from trainer_loop_worker import STEP0, nerfstudio_train_iteration, torch_optimizers
from two_rank_worker import build, shard
pipe, rb, batch, rnd = build("cuda:0", 1, 0)
rbs, bs, rs = shard(rb, batch, rnd, 1, 0, "cuda:0")
pipe.model.illumination_shard = (0, 1)
pipe.model.illumination_sampler.shared_across_ranks = True
pipe.exchange_with_one_rank()
pipe.config.graph_replay, pipe.config.graph_replay_warmup = True, 1
opts, scheds = torch_optimizers(pipe, fused=True)
losses = [float(nerfstudio_train_iteration(pipe, opts, scheds, STEP0 + i, ray_bundle=rbs, batch=bs, randoms=rs)) for i in range(4)]
torch.cuda.synchronize()
print("captured:", pipe._train_graph is not None, "losses", losses, flush=True)
dist.barrier(); dist.destroy_process_group()
