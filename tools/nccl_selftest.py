"""One-rank RCCL self-test of the calls bench.py makes when world > 1: init with device_id,
all_reduce MAX / SUM on device tensors, barrier, and the all_gather of proof records
(backend.gather_proofs on device tensors, [cap, 33] int64 per rank)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnark_crypto_primitives_amd import backend  # noqa: E402

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
p = torch.arange(1024 * 32, dtype=torch.int64, device=dev).reshape(1024, 32)
st = torch.zeros(1024, dtype=torch.int32, device=dev)
st[17] = -5
gp, gs = backend.gather_proofs(p, st, 1024, force=True)
torch.cuda.synchronize()
assert gp.device.type == "cuda" and torch.equal(gp, p) and torch.equal(gs, st)
hp, hs = backend.gather_proofs(p.cpu().numpy().view(np.uint64), st.cpu().numpy(), 1024, dev,
                               force=True)
assert np.array_equal(hp.view(np.int64), p.cpu().numpy()) and hs[17] == -5
print("nccl ok", float(t.item()), int(u.item()), "all_gather of", tuple(gp.shape), "proof records ok")
dist.destroy_process_group()
