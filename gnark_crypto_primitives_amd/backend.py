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


def gather_proofs(local_proofs, local_status, batch: int, device=None, force: bool = False):
    """all_gather of each rank's shard -> full [batch, 32] proofs and [batch] status on every rank.
    Shards may differ by one proof; they are padded to the largest shard for the collective.

    numpy in -> numpy out (staged through ``device``: "cpu" for gloo, a cuda device for RCCL);
    torch tensors in (int64 [n, 32] proofs, int32 [n] status, e.g. the buffers zkmi_prove_collect
    wrote on the GPU) -> torch tensors out on the same device, no host round trip.
    ``force``: run the collective even in a one-rank group (self-tests of the RCCL path)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return local_proofs, local_status
    world = dist.get_world_size()
    cap = -(-batch // world)
    as_tensor = isinstance(local_proofs, torch.Tensor)
    if as_tensor:
        dev = local_proofs.device
        lp, ls = local_proofs.view(torch.int64).reshape(-1, 32), local_status.to(torch.int64)
    else:
        dev = device if device is not None else "cpu"
        lp = torch.from_numpy(np.ascontiguousarray(local_proofs).view(np.int64).reshape(-1, 32)).to(dev)
        ls = torch.from_numpy(local_status.astype(np.int64)).to(dev)
    n = lp.shape[0]
    if n > cap:
        raise ValueError(f"shard of {n} proofs exceeds ceil({batch} / {world})")
    buf = torch.zeros((cap, 33), dtype=torch.int64, device=dev)
    if n:
        buf[:n, :32] = lp
        buf[:n, 32] = ls
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    parts_p, parts_s = [], []
    for r in range(world):
        lo, hi = shard_range(batch, r, world)
        parts_p.append(out[r][:hi - lo, :32])
        parts_s.append(out[r][:hi - lo, 32])
    proofs = torch.cat(parts_p)
    status = torch.cat(parts_s).to(torch.int32)
    if as_tensor:
        return proofs, status
    return (proofs.cpu().numpy().view(np.uint64).reshape(batch, 32),
            status.cpu().numpy().astype(np.int32))


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


def launch_local_ranks(argv, n_ranks: int, port: int, extra_env=None) -> int:
    """Start ``n_ranks`` fresh processes of ``argv`` on this node, one per GPU, with the
    torch.distributed environment a launcher would give them (RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR = 127.0.0.1, MASTER_PORT).  Children, never an exec: the caller keeps running (it
    must not have initialised HIP itself), forwards rank 0's stdout to its own stdout, sends the
    other ranks' stdout to stderr, and returns 0 or the first non-zero exit code -- in which case
    the ranks still running are terminated by PID (a dead rank would leave the others waiting in
    a collective for ever)."""
    import subprocess
    import sys
    import time
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks),
                   LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    try:
        out0 = procs[0].stdout
        os.set_blocking(out0.fileno(), False)
        live = set(range(n_ranks))
        while live:
            chunk = out0.read() if not out0.closed else None
            if chunk:
                sys.stdout.buffer.write(chunk)
                sys.stdout.buffer.flush()
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"launch_local_ranks: rank {r} exited with {code}; stopping the others",
                              file=sys.stderr, flush=True)
            if rc:
                break
            if live:
                time.sleep(0.05)
        if rc:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
        os.set_blocking(out0.fileno(), True)
        rest = out0.read()
        if rest:
            sys.stdout.buffer.write(rest)
            sys.stdout.buffer.flush()
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
