"""GPU PLONK prover (zkmi_plonk_round1..5 through plonk.Prover) against the CPU restatement
(oracle/plonk_ref.py): same SRS, same blinding, same transcript -> the same nine commitments and
six evaluations, bit for bit; every proof verifies with the product verifier AND the oracle's;
tampering / wrong public inputs are rejected; an unsatisfied witness is flagged by the solver.
Parity unpinned with respect to gnark (the reference holds no PLONK vector)."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, plonk
from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
from gnark_crypto_primitives_amd.frontend.scs import compile_scs
from gnark_crypto_primitives_amd.hash import poseidon_native
from gnark_crypto_primitives_amd.tree import smt_witness

pytestmark = pytest.mark.gpu
R = plonk.R


def _oracle_proof(P, key, sc, inp, blind, python_loops=False):
    """the CPU restatement's proof: plonk_ref.prove_fast (loops in C; tests/test_plonk.py pins it to
    the plain-integer loops of plonk_ref.prove), or those Python loops themselves"""
    _, a, b, c = sc.run_vprogram(inp)
    fn = P.prove if python_loops else P.prove_fast
    return fn(key, a, b, c, inp[:sc.n_public - 1], blind)


def _same(gp, op):
    return all(getattr(gp, f) == op[f] for f in plonk.Proof.FIELDS) and gp.ev == op["ev"]


@pytest.mark.parametrize("wbits", [0, 7, 310])
def test_poseidon_plonk_vs_oracle(zk_ctx, wbits):
    from oracle import plonk_ref as P
    sc = compile_scs(circuits.PoseidonCircuit())
    pk = plonk.setup(zk_ctx, sc, 7)
    key = P.setup(sc, 7)
    assert pk.com == key["com"] and pk.g2_tau == key["g2_tau"]          # same preprocessed key
    prover = plonk.Prover(zk_ctx, sc, pk, window_bits=wbits, max_batch=128)
    rng = random.Random(5)
    batch = 67                                                            # ragged second wavefront
    datas = [rng.randrange(R) for _ in range(batch)]
    inps = [sc.assignment_vector({"Data": d, "Hash": poseidon_native.hash([d])}) for d in datas]
    inps[3] = sc.assignment_vector({"Data": datas[3], "Hash": 77})        # unsatisfied
    blinds = [[rng.randrange(R) for _ in range(9)] for _ in range(batch)]
    blinds[1] = [0] * 9                                                   # no blinding at all
    proofs, status = prover.prove(np.stack([to_mont_array(v) for v in inps]),
                                  np.stack([to_mont_array(v) for v in blinds]))
    prover.close()
    assert list(status != 0) == [i == 3 for i in range(batch)]
    for i in (0, 1, 2, 63, 64, 66):
        want = _oracle_proof(P, key, sc, inps[i], blinds[i], python_loops=(i == 1))
        assert _same(proofs[i], want), i
    pub = inps[0][:1]
    assert plonk.verify(pk, pub, proofs[0])
    assert P.verify(key, pub, {**{f: getattr(proofs[0], f) for f in plonk.Proof.FIELDS},
                               "ev": proofs[0].ev})
    assert not plonk.verify(pk, [(pub[0] + 1) % R], proofs[0])
    bad = plonk.Proof(**{**{f: getattr(proofs[0], f) for f in plonk.Proof.FIELDS},
                         "ev": (proofs[0].ev[0] + 1,) + proofs[0].ev[1:]})
    assert not plonk.verify(pk, pub, bad)
    assert not plonk.verify(pk, inps[3][:1], proofs[3])                   # the unsatisfied witness


def test_plonk_lagrange_basis_commitments(zk_ctx):
    """Wire columns committed in the Lagrange basis (zkmi_plonk_pk_desc.lag_*: the witness values
    are the scalars, blinding through the two extra points) give the commitments of the
    coefficient-form path and of the oracle, bit for bit; forced on here, the address circuit's
    full-size test below runs it by default."""
    from oracle import plonk_ref as P
    sc = compile_scs(circuits.smt_inclusion_circuit(4))
    pk = plonk.setup(zk_ctx, sc, 13)
    key = P.setup(sc, 13)
    rng = random.Random(21)
    inps = [sc.assignment_vector(smt_witness.synthetic_inclusion(rng, 4, k % 4)) for k in range(70)]
    blinds = [[rng.randrange(R) for _ in range(9)] for _ in inps]
    blinds[2] = [0] * 9
    args = (np.stack([to_mont_array(v) for v in inps]), np.stack([to_mont_array(v) for v in blinds]))
    out = {}
    for mode in (True, False):
        prover = plonk.Prover(zk_ctx, sc, pk, max_batch=128, lagrange=mode)
        assert prover.lagrange == mode
        out[mode] = prover.prove(*args)
        prover.close()
    assert not out[True][1].any() and not out[False][1].any()
    for i in range(70):
        assert out[True][0][i] == out[False][0][i], i
    for i in (0, 2, 63, 64, 69):
        assert _same(out[True][0][i], _oracle_proof(P, key, sc, inps[i], blinds[i])), i
        assert plonk.verify(pk, [], out[True][0][i])


def test_smt_plonk_vs_oracle(zk_ctx):
    """SMT inclusion verifier, 8 levels: 10 264 gates, domain 2^14 (quotient on 2^16)."""
    from oracle import plonk_ref as P
    sc = compile_scs(circuits.smt_inclusion_circuit(8))
    assert sc.log_n == 14
    pk = plonk.setup(zk_ctx, sc, 9)
    key = P.setup(sc, 9)
    assert pk.com == key["com"]
    prover = plonk.Prover(zk_ctx, sc, pk, max_batch=64)
    rng = random.Random(8)
    inps = [sc.assignment_vector(smt_witness.synthetic_inclusion(rng, 8, k)) for k in (0, 3, 7)]
    blinds = [[rng.randrange(R) for _ in range(9)] for _ in inps]
    proofs, status = prover.prove(np.stack([to_mont_array(v) for v in inps]),
                                  np.stack([to_mont_array(v) for v in blinds]))
    prover.close()
    assert not status.any()
    assert _same(proofs[1], _oracle_proof(P, key, sc, inps[1], blinds[1]))
    for i in range(3):
        assert plonk.verify(pk, [], proofs[i])


def test_config5_address_on_plonk_backend(zk_ctx):
    """BASELINE config 5 as it is worded: the secp256k1 address circuit (`ecdsa.DeriveAddress`,
    ecc/secp256k1/ecdsa/address.go:14-40) on the PLONK backend at full size: 231 270 gates, domain
    2^18, quotient on 2^20.  (i) two complete proofs -- nine commitments, six evaluations -- bit
    for bit against the CPU oracle (its loops in C at this size), (ii) the oracle's verifier and the
    product's on the complete proofs, (iii) a wrong address flagged by the solver and rejected by
    the verifiers.  Parity unpinned with respect to gnark (address_test.go:57 proves with Groth16)."""
    from oracle import plonk_ref as P
    from gnark_crypto_primitives_amd import workloads
    from tests import helpers as H
    _, gen, _ = workloads.build("address")
    sc = H.compiled("address-scs")
    assert sc.log_n == 18 and sc.n_gates > 200_000
    pk = plonk.setup(zk_ctx, sc, 11)
    prover = plonk.Prover(zk_ctx, sc, pk, max_batch=64)
    rng = random.Random(55)
    asg = [gen(rng) for _ in range(3)]
    wrong = dict(asg[2])
    wrong["Address"] = (wrong["Address"] + 1) % (1 << 160)
    inps = [sc.assignment_vector(a) for a in (asg[0], asg[1], wrong)]
    blinds = [[rng.randrange(R) for _ in range(9)] for _ in inps]
    proofs, status = prover.prove(np.stack([to_mont_array(v) for v in inps]),
                                  np.stack([to_mont_array(v) for v in blinds]))
    prover.close()
    assert list(status != 0) == [False, False, True]
    n_pub = sc.n_public - 1
    # (i) ALL nine commitments and six evaluations of two proofs against the oracle (the C twin of
    # plonk_ref.prove, oracle/c/zkref_plonk.inc; tests/test_plonk.py pins it to the Python loops)
    okey = P.setup_fast(sc, pk.srs_g1, pk.com, pk.g2_tau)
    for i in (0, 1):
        _, a, b, c = sc.run_vprogram(inps[i])
        want = P.prove_fast(okey, a, b, c, inps[i][:n_pub], blinds[i])
        assert _same(proofs[i], want), i
        if i == 1:
            assert P.round1_commitments(pk.srs_g1, sc.log_n, a, b, c, blinds[1]) == \
                (want["a"], want["b"], want["c"])
    # (ii) complete proofs under both verifiers (the oracle's takes the verifying key as data)
    vkey = {"log_n": pk.log_n, "n_pub": n_pub, "com": pk.com, "g2_tau": pk.g2_tau}
    for i in (0, 1):
        pub = inps[i][:n_pub]
        assert plonk.verify(pk, pub, proofs[i])
        assert P.verify(vkey, pub, {**{f: getattr(proofs[i], f) for f in plonk.Proof.FIELDS},
                                    "ev": proofs[i].ev})
    # (iii) the wrong address: neither its own public inputs nor the right ones make it verify
    assert not plonk.verify(pk, inps[2][:n_pub], proofs[2])
    assert not plonk.verify(pk, inps[1][:n_pub], proofs[0])


def test_plonk_prove_in_one_call_equals_the_round_by_round_prover(zk_ctx):
    """zkmi_plonk_prove (transcript hashed in C++ between the rounds) against plonk.Prover.prove (the
    same rounds driven from Python with the Python transcript): identical proofs, on the Poseidon
    circuit (one public input) and on a circuit with three public inputs."""
    from tests.test_frontend import Mixed, _mixed_expected
    rng = random.Random(17)
    sc = compile_scs(circuits.PoseidonCircuit())
    pk = plonk.setup(zk_ctx, sc, 7)
    prover = plonk.Prover(zk_ctx, sc, pk, max_batch=128)
    datas = [rng.randrange(R) for _ in range(70)]
    inps = [sc.assignment_vector({"Data": d, "Hash": poseidon_native.hash([d])}) for d in datas]
    inps[5] = sc.assignment_vector({"Data": 1, "Hash": 2})
    inp = np.stack([to_mont_array(v) for v in inps])
    blind = np.stack([to_mont_array([rng.randrange(R) for _ in range(9)]) for _ in inps])
    want, wstatus = prover.prove(inp, blind)
    rec, status = prover.prove_raw(inp, blind)
    got = prover.proofs_of(rec)
    prover.close()
    assert np.array_equal(status, wstatus) and list(np.nonzero(status)[0]) == [5]
    assert all(got[i] == want[i] for i in range(70) if i != 5)
    assert plonk.verify(pk, inps[0][:1], got[0])
    sc = compile_scs(Mixed())
    pk = plonk.setup(zk_ctx, sc, 8)
    prover = plonk.Prover(zk_ctx, sc, pk, max_batch=64)
    asg = [(7 + i, 1000 + 3 * i) for i in range(5)]
    inps = [sc.assignment_vector({"X": x, "Y": y, "Z": _mixed_expected(x, y)}) for x, y in asg]
    inp = np.stack([to_mont_array(v) for v in inps])
    blind = np.stack([to_mont_array([rng.randrange(R) for _ in range(9)]) for _ in inps])
    want, _ = prover.prove(inp, blind)
    got = prover.proofs_of(prover.prove_raw(inp, blind)[0])
    prover.close()
    assert got == want and plonk.verify(pk, inps[2][:sc.n_public - 1], got[2])
