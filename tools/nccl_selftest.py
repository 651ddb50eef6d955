"""One-rank RCCL self-test of the calls bench.py makes when world > 1 (init with device_id,
all_reduce MAX / SUM on device tensors, barrier)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29513")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([3], dtype=torch.int64, device=dev)
dist.all_reduce(u)
dist.barrier()
torch.cuda.synchronize()
print("nccl ok", float(t.item()), int(u.item()))
dist.destroy_process_group()
