"""N > 1 path: contiguous batch split + all_gather of proof records, world_size 2 over gloo.
The per-rank prover is the CPU oracle here (the GPU path needs a GPU); the sharding and gather
logic under test is the code bench.py and production use."""
import os
import socket

import numpy as np
import pytest

from gnark_crypto_primitives_amd import backend


def test_shard_range_partitions():
    for batch in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [backend.shard_range(batch, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import random

        from gnark_crypto_primitives_amd import circuits, groth16
        from gnark_crypto_primitives_amd.frontend import compile_circuit
        from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
        from gnark_crypto_primitives_amd.hash import poseidon_native
        from oracle import cref
        from tests import helpers as H
        cc = compile_circuit(circuits.PoseidonCircuit())
        mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
        pk, _, _ = groth16.setup(cc, 3, mul)          # same seed on every rank: replicated key
        rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
        rng = random.Random(0)
        batch = 5                                      # odd: shards of 3 and 2
        datas = [rng.randrange(H.R) for _ in range(batch)]
        inp = np.stack([to_mont_array(cc.assignment_vector(
            {"Data": d, "Hash": poseidon_native.hash([d])})) for d in datas])
        inp[4, 0, 0] ^= 1                              # one unsatisfied witness, on rank 1
        rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)])
                       for _ in range(batch)])

        def prover(i, r):
            p, s, _ = cref.groth16_prove_batch(rh, ph, i, r, 1)
            return p, s
        proofs, status = backend.prove_sharded(prover, inp, rs)
        want, wstatus = prover(inp, rs)
        ok = wstatus == 0
        good = bool(np.array_equal(status != 0, wstatus != 0)
                    and np.array_equal(proofs[ok], want[ok])
                    and list(wstatus != 0) == [False] * 4 + [True])
        # the form bench.py uses: this rank's shard as torch tensors (as zkmi_prove_collect leaves
        # them), gathered without a numpy round trip -- strong scaling: the global batch is split
        import torch
        lo, hi = backend.shard_range(batch, rank, world)
        tp = torch.from_numpy(want[lo:hi].view(np.int64).copy())
        ts = torch.from_numpy(wstatus[lo:hi].astype(np.int32))
        gp, gs = backend.gather_proofs(tp, ts, batch)
        good = good and isinstance(gp, torch.Tensor) and gp.shape == (batch, 32) \
            and np.array_equal(gp.numpy().view(np.uint64), want) \
            and np.array_equal(gs.numpy(), wstatus)
        q.put((rank, good))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_prove_sharded_world2_gloo():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_gather_proofs_forced_single_rank():
    """world = 1 with force=True still runs the collective (what tools/nccl_selftest.py does over
    RCCL on the GPU box)."""
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        p = np.arange(7 * 32, dtype=np.uint64).reshape(7, 32)
        st = np.array([0, -5, 0, 0, 0, 0, -5], dtype=np.int32)
        gp, gs = backend.gather_proofs(p, st, 7, force=True)
        assert np.array_equal(gp, p) and np.array_equal(gs, st)
        tp, ts = backend.gather_proofs(torch.from_numpy(p.view(np.int64)), torch.from_numpy(st), 7,
                                       force=True)
        assert np.array_equal(tp.numpy().view(np.uint64), p) and np.array_equal(ts.numpy(), st)
    finally:
        dist.destroy_process_group()
