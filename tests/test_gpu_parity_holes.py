"""Parity holes named by the round-2 review:
(a) every template instance of the witness solver -- solve_vliw_kernel<S>, S = lanes_per_proof in
    {1, 2, 4, 8, 16, 32, 64} -- against the oracle's constraint-by-constraint solver (wires, a, b, c);
(b) BASELINE config 3 at full size: smt_verifier_circuit(160), inclusion + exclusion,
    isOld0 in {0, 1}, 1024 proofs under the auto plan, 64-lane oracle sample
    (tree/smt/verifier.go:66-81,102);
(c) BASELINE config 4 at full size: ElGamal homomorphic add x 8192, 64-lane oracle sample
    (elgamal/ciphertext.go:24-32);
(d) two Arbo-160 keys on one GPU under a 64 GB table budget each, both proving oracle-exact samples.
"""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
from gnark_crypto_primitives_amd.tree import smt_witness
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64])
def test_solver_every_lane_count_vs_oracle(zk_ctx, lanes):
    from oracle import cref
    from tests.test_frontend import Mixed, _mixed_expected
    rng = random.Random(100 + lanes)
    cases = []
    cc = compile_circuit(Mixed(), lanes)
    asg = []
    for i in range(70):                                   # two wavefronts at S = 1, ragged
        x = rng.randrange(1 << 16)
        y = x if i % 5 == 0 else rng.randrange(H.R)
        asg.append({"X": x, "Y": y, "Z": _mixed_expected(x, y)})
    asg[3]["Z"] = (asg[3]["Z"] + 1) % H.R
    asg[64] = {"X": 5, "Y": H.R - 2, "Z": 0}
    cases.append((cc, asg, {3, 64}))
    cc = compile_circuit(circuits.smt_inclusion_circuit(8), lanes)
    asg = [smt_witness.synthetic_inclusion(rng, 8, 1 + i % 7) for i in range(70)]
    asg[69] = dict(asg[69], Root=(asg[69]["Root"] + 1) % H.R)
    cases.append((cc, asg, {69}))
    for cc, asg, bad in cases:
        assert cc.lanes_per_proof == lanes
        pk, _, _ = groth16.setup(cc, 7, groth16.gpu_mul(zk_ctx))
        prover = groth16.Prover(zk_ctx, cc, pk, 7, 5)
        inp = np.stack([to_mont_array(cc.assignment_vector(a)) for a in asg])
        status, wires, abc = prover.solve(inp, want_wires=True, want_abc=True)
        rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in asg])
        proofs, pstatus = prover.prove(inp, rs)
        prover.close()
        assert set(np.nonzero(status)[0]) == bad and np.array_equal(status, pstatus)
        rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
        for i in range(len(asg)):
            st, w, a, b, c = cref.r1cs_solve(rh, inp[i])
            assert (st != 0) == (i in bad), i
            if i in bad:
                continue
            assert np.array_equal(wires[i], w), (lanes, i)
            assert np.array_equal(abc[0][i], a) and np.array_equal(abc[1][i], b) \
                and np.array_equal(abc[2][i], c), (lanes, i)
        want, wstatus, _ = cref.groth16_prove_batch(rh, ph, inp, rs)
        ok = status == 0
        assert np.array_equal(proofs[ok], want[ok])


def _sample(rng, B, extra=()):
    return sorted(set([0, 1, 63, 64, 65, 511, 512, B - 2, B - 1] + list(extra) +
                      [rng.randrange(B) for _ in range(64)]))[:64]


def test_config3_verifier160_full_size_auto_plan(zk_ctx):
    from oracle import cref
    levels, B = 160, 1024
    cc = compile_circuit(circuits.smt_verifier_circuit(levels))
    pk, _, _ = groth16.setup(cc, 3, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, 0, 0)
    info = zk_ctx.pk_info(prover.pk_h)
    assert info["g1_comb_k"] >= 16 and info["g2_comb_k"] >= 16, info
    rng = random.Random(303)
    distinct = []
    for i in range(192):
        populated = 1 + (i * 7) % 158
        w = smt_witness.synthetic_inclusion(rng, levels, populated)
        kind = i % 4
        if kind == 0:                                   # inclusion
            a = dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Fnc=0)
        elif kind in (1, 2):                            # exclusion next to an existing leaf
            mask = (1 << populated) - 1
            other = (w["Key"] & mask) | (((w["Key"] >> populated) ^ 1) << populated)
            a = dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Key=other, Value=0, Fnc=1)
        else:                                           # exclusion into an empty branch: isOld0 = 1
            a = smt_witness.synthetic_exclusion_empty(rng, levels, populated)
        distinct.append(to_mont_array(cc.assignment_vector(a)))
    try:
        order = [rng.randrange(len(distinct)) for _ in range(B)]
        inp = np.stack([distinct[i] for i in order])
        bad = 321
        inp[bad, 0, 0] ^= np.uint64(1)
        rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in range(B)])
        proofs, status = prover.prove(inp, rs)
        assert list(np.nonzero(status)[0]) == [bad]
        sample = [i for i in _sample(rng, B, (bad - 1, bad + 1)) if i != bad]
        kinds = {order[i] % 4 for i in sample}
        assert kinds == {0, 1, 2, 3}                    # inclusion, exclusion, isOld0 = 1 all sampled
        want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk),
                                                    inp[sample], rs[sample], 16)
        assert not wstatus.any() and np.array_equal(proofs[sample], want)
    finally:
        prover.close()


def test_config4_elgamal_add_8192(zk_ctx):
    from gnark_crypto_primitives_amd import workloads
    from oracle import cref
    B = 8192
    circuit, gen, _ = workloads.build("elgamal-add")
    cc = compile_circuit(circuit)
    pk, _, _ = groth16.setup(cc, 4, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, 0, 0, max_batch=B)
    rng = random.Random(404)
    distinct = [to_mont_array(cc.assignment_vector(gen(rng))) for _ in range(20)]   # 0.3 s each (Python curve arithmetic)
    try:
        order = [rng.randrange(len(distinct)) for _ in range(B)]
        inp = np.stack([distinct[i] for i in order])
        bad = 4097
        inp[bad, 8, 0] ^= np.uint64(1)                   # Sum.C1.X
        rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in range(B)])
        proofs, status = prover.prove(inp, rs)
        assert list(np.nonzero(status)[0]) == [bad]
        sample = [i for i in _sample(rng, B, (4095, 4096, 4098, 8127, 8128)) if i != bad]
        want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk),
                                                    inp[sample], rs[sample], 16)
        assert not wstatus.any() and np.array_equal(proofs[sample], want)
    finally:
        prover.close()


def test_two_arbo160_keys_share_one_gpu_at_64gb_each(zk_ctx):
    """The bounded-memory operating point: two independent Arbo-160 keys (different trapdoors) with
    table_budget_bytes = 64 GB each live on one context and both prove oracle-exact samples."""
    from oracle import cref
    B = 1024
    cc = H.compiled("arbo160")
    rng = random.Random(64)
    ws = [to_mont_array(cc.assignment_vector(smt_witness.synthetic_inclusion(rng, 160, 1 + i % 159)))
          for i in range(128)]
    provers = []
    try:
        for seed in (21, 22):
            pk, _, _ = groth16.setup(cc, seed, groth16.gpu_mul(zk_ctx))
            p = groth16.Prover(zk_ctx, cc, pk, 0, 0, table_budget_bytes=64 * 10 ** 9)
            info = zk_ctx.pk_info(p.pk_h)
            assert info["g1_table_bytes"] + info["g2_table_bytes"] <= 64 * 10 ** 9, info
            assert info["g1_comb_k"] >= 12, info
            provers.append((p, pk, info))
        for k, (p, pk, info) in enumerate(provers):
            inp = np.stack([ws[rng.randrange(len(ws))] for _ in range(B)])
            rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in range(B)])
            # interleave: submit on one key while the other key's batch is collected
            p.submit(inp, rs)
            proofs, status = p.collect()
            assert not status.any()
            sample = _sample(rng, B)[::2]                      # 32 lanes per key
            want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk),
                                                        inp[sample], rs[sample], 16)
            assert not wstatus.any() and np.array_equal(proofs[sample], want), k
    finally:
        for p, _, _ in provers:
            p.close()
