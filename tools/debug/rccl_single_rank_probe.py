"""One-GPU sanity probe of the RCCL plumbing bench.py --gpus N relies on: a world-size-1 "nccl" group (RCCL cannot place
two ranks on one GPU, so the p2p hand-off itself needs a multi-GPU node) — library load, communicator init with device_id,
float64 / int64 collectives, barrier, teardown.  Prints one line."""
import os

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.arange(4, device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = [torch.zeros(5, device=dev, dtype=torch.float64)]
dist.all_gather(g, torch.ones(5, device=dev, dtype=torch.float64))
dist.barrier()
torch.cuda.synchronize()
print("rccl single-rank ok:", dist.get_backend(), t.tolist(), g[0].tolist(), torch.cuda.nccl.version())
dist.destroy_process_group()
