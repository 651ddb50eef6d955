"""Groth16 commitment extension on the GPU: circuits built with api.Commit /
std/rangecheck are solved in phases (the solver stops at the commitment, the Pedersen MSM and the
hash supply the challenge), proved, and compared bit for bit -- proof, commitments, proof of
knowledge -- with the C oracle; the product's verifier accepts them.  Reference path:
ecc/secp256k1/ecdsa/address.go:14-40 -> utils/uints.go:14-28 (uints.New -> rangecheck).
Parity unpinned (see tests/test_commitment.py)."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16, lib, verify
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
from tests import helpers as H
from tests.test_commitment import RangeCircuit, TwoCommitments

pytestmark = pytest.mark.gpu


def _check(zk_ctx, cc, asg, bad, seed, wbits=(7, 5), publics=None, **plan):
    from oracle import cref
    pk, vk, _ = groth16.setup(cc, seed, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, *wbits, **plan)
    rng = random.Random(seed)
    inp = np.stack([to_mont_array(cc.assignment_vector(a)) for a in asg])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in asg])
    rh, ph, ch = cref.R1csHandle(cc), cref.PkHandle(pk), cref.CommitKeysHandle(pk)
    want, wcoms, wpoks, wstatus, _ = cref.groth16_prove_batch_ex(rh, ph, ch, inp, rs, 16)
    try:
        # twice, pipelined two deep: the second submit runs its phases beside the first batch's MSMs
        prover.submit(inp, rs)
        prover.submit(inp, rs)
        for _ in range(2):
            proofs, status, coms = prover.collect()
            assert set(np.nonzero(status)[0]) == set(bad) == set(np.nonzero(wstatus)[0])
            ok = status == 0
            assert np.array_equal(proofs[ok], want[ok])
            assert np.array_equal(coms[ok][:, :-1], wcoms[ok])
            assert np.array_equal(coms[ok][:, -1], wpoks[ok])
        if publics is not None:
            for i in list(np.nonzero(ok)[0])[:3]:
                assert verify.verify(vk, publics[i], proofs[i], coms[i, :-1], coms[i, -1])
                assert not verify.verify(vk, [publics[i][0] + 1] + publics[i][1:], proofs[i],
                                         coms[i, :-1], coms[i, -1])
        # the gnark drop-in entry with committed wires: solved wire vectors (commitment wires
        # included, from the oracle's gnark-style solver) through zkmi_prove_witness_submit
        good = [i for i in range(len(asg)) if i not in bad]
        W = np.stack([cref.r1cs_solve_ex(rh, ch, inp[i])[1] for i in good])
        prover.submit_witness(W, np.ascontiguousarray(rs[good]))
        p2, s2, c2 = prover.collect()
        assert not s2.any() and np.array_equal(p2, want[good])
        assert np.array_equal(c2[:, :-1], wcoms[good]) and np.array_equal(c2[:, -1], wpoks[good])
        with pytest.raises(lib.ZkmiError):      # the plain collect refuses a key with commitments
            prover.ctx.prove_submit(prover.pk_h, prover.cs_h, inp, len(asg), rs)
            try:
                prover.ctx.prove_collect(np.zeros((len(asg), 32), np.uint64),
                                         np.zeros(len(asg), np.int32))
            finally:
                prover.ctx.prove_collect(np.zeros((len(asg), 32), np.uint64),
                                         np.zeros(len(asg), np.int32),
                                         np.zeros((len(asg), prover.n_commitments + 1, 8), np.uint64))
    finally:
        prover.close()


@pytest.mark.parametrize("lanes", [1, 4, 16, 64])
def test_range_circuit(zk_ctx, lanes):
    cc = compile_circuit(RangeCircuit(), lanes)
    rng = random.Random(lanes)
    asg, bad = [], []
    for i in range(70):
        a, b = rng.randrange(1, 90), rng.randrange(1, 90)
        y = a | b << 8 | rng.getrandbits(48) << 16
        asg.append({"X": a * b, "Y": y})
    asg[5]["Y"] += 1 << 64          # a ninth byte: recomposition fails
    asg[64] = {"X": 8190, "Y": 0x2ac3}     # X + 3 >= 2^13
    asg[69]["X"] += 1
    bad = [5, 64, 69]
    _check(zk_ctx, cc, asg, bad, 31 + lanes, publics=[[a["X"]] for a in asg])


def test_two_commitments(zk_ctx):
    cc = compile_circuit(TwoCommitments())
    asg = [{"X": y * y % H.R, "Y": y} for y in range(3, 73)]
    asg[10]["X"] += 1
    _check(zk_ctx, cc, asg, [10], 41, publics=[[a["X"]] for a in asg])
    _check(zk_ctx, cc, asg, [10], 42, wbits=(0, 0))


def test_config5_address_with_commitment_range_checks(zk_ctx):
    """The address circuit with its 64 input bytes range-checked as gnark's uints.New does it for an
    R1CS builder (lookup + commitment, 144 committed wires), 2^18 domain, auto plan."""
    from gnark_crypto_primitives_amd.std.emulated import limbs_of
    from oracle import pyref
    cc = H.compiled("address-commit")
    assert cc.domain_log2() == 18 and len(cc.commitments) == 1
    assert len(cc.commitments[0]["private"]) == 144           # 128 nibbles + 16 multiplicities
    rng = random.Random(56)
    asg = []
    for priv in (1, 2, rng.randrange(1, pyref.SECP_N), rng.randrange(1, pyref.SECP_N),
                 rng.randrange(1, pyref.SECP_N)):
        pub = pyref.secp256k1_mul(priv)
        asg.append({"Address": pyref.eth_address(pub), "X": limbs_of(pub[0]),
                    "Y": limbs_of(pub[1])})
    assert asg[0]["Address"] == 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF
    asg[3] = dict(asg[3], Address=asg[2]["Address"])          # wrong address
    x = list(asg[4]["X"])
    x[1] += 1 << 64                                            # a limb byte out of range
    asg[4] = dict(asg[4], X=x)
    _check(zk_ctx, cc, asg, [3, 4], 57, wbits=(0, 0), publics=[[a["Address"]] for a in asg],
           max_batch=64)


def test_config5_address_as_gnark_compiles_it(zk_ctx):
    """ecdsa.DeriveAddress with everything on bytes, as gnark's uints / sha3 build it: 217 366
    constraints, 378 133 wires, 174 272 of them behind one commitment, domain 2^18.  Three proofs
    (public vectors for keys 1 and 2), one with a wrong address; proof, commitment and proof of
    knowledge bit-exact against the C oracle, verifying under the product's verifier."""
    from gnark_crypto_primitives_amd.std.emulated import limbs_of
    from oracle import pyref
    cc = H.compiled("address-bytes")
    assert cc.domain_log2() == 18 and len(cc.commitments) == 1
    asg = []
    for priv in (1, 2, 0xC0FFEE):
        pub = pyref.secp256k1_mul(priv)
        asg.append({"Address": pyref.eth_address(pub), "X": limbs_of(pub[0]), "Y": limbs_of(pub[1])})
    assert asg[0]["Address"] == 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF
    asg[2] = dict(asg[2], Address=asg[1]["Address"])
    _check(zk_ctx, cc, asg, [2], 58, wbits=(0, 0), publics=[[a["Address"]] for a in asg],
           max_batch=64)
