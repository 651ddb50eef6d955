"""Batch sharding across the GPUs of one node: one process per GPU, no data-path collective.

Every proof depends only on the (replicated, read-only) proving key and constraint system and on
its own inputs, so a batch of B proofs splits into contiguous shards [g*B/G, (g+1)*B/G) with no
exchange during compute.  torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests) is used only to gather the fixed-size proof records.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(batch: int, rank: int, world: int):
    """Contiguous split; the first (batch % world) ranks take one extra proof."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), \
        int(os.environ.get("LOCAL_RANK", 0))


def gather_proofs(local_proofs: np.ndarray, local_status: np.ndarray, batch: int, device=None):
    """all_gather of each rank's shard -> full [batch, 32] proofs and [batch] status on every rank.
    Shards may differ by one proof; they are padded to the largest shard for the collective."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_proofs, local_status
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = -(-batch // world)
    dev = device if device is not None else "cpu"
    buf = torch.zeros((cap, 33), dtype=torch.int64, device=dev)
    n = local_proofs.shape[0]
    if n:
        buf[:n, :32] = torch.from_numpy(local_proofs.view(np.int64)).to(dev)
        buf[:n, 32] = torch.from_numpy(local_status.astype(np.int64)).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    proofs = np.zeros((batch, 32), dtype=np.uint64)
    status = np.zeros(batch, dtype=np.int32)
    for r in range(world):
        lo, hi = shard_range(batch, r, world)
        arr = out[r].cpu().numpy()
        proofs[lo:hi] = arr[:hi - lo, :32].view(np.uint64)
        status[lo:hi] = arr[:hi - lo, 32].astype(np.int32)
    return proofs, status


def prove_sharded(prover_fn, inputs: np.ndarray, rs: np.ndarray, device=None):
    """prover_fn(inputs_shard, rs_shard) -> (proofs, status) on this rank's GPU; returns the
    gathered result for the whole batch."""
    import torch.distributed as dist
    batch = inputs.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    lo, hi = shard_range(batch, rank, world)
    if hi > lo:
        proofs, status = prover_fn(np.ascontiguousarray(inputs[lo:hi]),
                                   np.ascontiguousarray(rs[lo:hi]))
    else:
        proofs, status = np.zeros((0, 32), np.uint64), np.zeros(0, np.int32)
    return gather_proofs(proofs, status, batch, device)
