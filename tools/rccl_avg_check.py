import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.arange(8, device="cuda:0", dtype=torch.float32)
dist.all_reduce(t, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
print("AVG ok", t.tolist(), dist.get_backend())
dist.destroy_process_group()
