"""PLONK lowering (frontend/scs.py) and the CPU restatement of the prover / verifier
(oracle/plonk_ref.py): gate + copy constraints hold on solved witnesses, invalid witnesses break
them, oracle proofs verify, tampering is rejected.  Parity unpinned (no PLONK vector exists in the
reference): these pin the protocol the GPU path implements (tests/test_gpu_plonk.py)."""
import random

import pytest

from gnark_crypto_primitives_amd import circuits
from gnark_crypto_primitives_amd.frontend.scs import compile_scs
from gnark_crypto_primitives_amd.hash import poseidon_native
from oracle import plonk_ref as P
from tests.test_frontend import Mixed, _mixed_expected

R = P.R


def test_scs_lowering_gates_and_copies():
    sc = compile_scs(Mixed())
    rng = random.Random(1)
    for i in range(4):
        x = rng.randrange(1 << 16)
        y = x if i == 0 else rng.randrange(R)
        inp = sc.assignment_vector({"X": x, "Y": y, "Z": _mixed_expected(x, y)})
        _, a, b, c = sc.run_vprogram(inp)
        assert sc.last_status == 0
        assert sc.is_satisfied(a, b, c, inp[:sc.n_public - 1]) == (True, -1)
        assert None not in a and None not in b and None not in c       # every gate row emitted
    bad = sc.assignment_vector({"X": 3, "Y": 4, "Z": 5})
    _, a, b, c = sc.run_vprogram(bad)
    assert sc.last_status == -5 and not sc.is_satisfied(a, b, c, bad[:sc.n_public - 1])[0]
    # a wire value changed in one column only violates a copy constraint or a gate
    inp = sc.assignment_vector({"X": 9, "Y": 11, "Z": _mixed_expected(9, 11)})
    _, a, b, c = sc.run_vprogram(inp)
    c2 = list(c)
    c2[len(c2) // 2] = (c2[len(c2) // 2] + 1) % R
    assert not sc.is_satisfied(a, b, c2, inp[:sc.n_public - 1])[0]


def test_plonk_oracle_proves_and_verifies():
    sc = compile_scs(circuits.PoseidonCircuit())
    assert sc.log_n == 10
    key = P.setup(sc, 7)
    rng = random.Random(3)
    d = 12345
    inp = sc.assignment_vector({"Data": d, "Hash": poseidon_native.hash([d])})
    _, a, b, c = sc.run_vprogram(inp)
    pub = inp[:sc.n_public - 1]
    blind = [rng.randrange(R) for _ in range(9)]
    proof = P.prove(key, a, b, c, pub, blind)
    assert P.verify(key, pub, proof)
    assert P.prove(key, a, b, c, pub, blind) == proof                      # deterministic
    # the first-round helper the full-size GPU test uses is the prover's own first round
    assert P.round1_commitments(key["srs"], sc.log_n, a, b, c, blind) == \
        (proof["a"], proof["b"], proof["c"])
    assert P.prove(key, a, b, c, pub, [x + 1 for x in blind]) != proof     # blinding matters
    bad = dict(proof, ev=(proof["ev"][0] + 1,) + proof["ev"][1:])
    assert not P.verify(key, pub, bad)
    assert not P.verify(key, [(pub[0] + 1) % R], proof)
    bad = dict(proof, wz=proof["wzw"])
    assert not P.verify(key, pub, bad)
    # an unsatisfied witness has no quotient polynomial
    inp2 = sc.assignment_vector({"Data": d, "Hash": 5})
    _, a, b, c = sc.run_vprogram(inp2)
    with pytest.raises(AssertionError):
        P.prove(key, a, b, c, inp2[:1], blind)


def test_fast_oracle_prover_equals_the_python_loops():
    """oracle/c/zkref_plonk.inc (grand product, quotient, evaluations, divisions in C) against the
    plain-integer loops of plonk_ref.prove: same nine commitments and six evaluations, on the
    Poseidon circuit and on the every-opcode circuit; an unsatisfied witness is refused."""
    rng = random.Random(8)
    sc = compile_scs(circuits.PoseidonCircuit())
    key = P.setup(sc, 7)
    for d in (12345, 0, R - 1):
        inp = sc.assignment_vector({"Data": d, "Hash": poseidon_native.hash([d])})
        _, a, b, c = sc.run_vprogram(inp)
        pub = inp[:sc.n_public - 1]
        blind = [rng.randrange(R) for _ in range(9)]
        fast = P.prove_fast(key, a, b, c, pub, blind)
        assert fast == P.prove(key, a, b, c, pub, blind)
        assert P.verify(key, pub, fast)
    inp2 = sc.assignment_vector({"Data": 1, "Hash": 5})
    _, a, b, c = sc.run_vprogram(inp2)
    with pytest.raises(AssertionError):
        P.prove_fast(key, a, b, c, inp2[:1], blind)
    sc = compile_scs(Mixed())
    key = P.setup(sc, 9)
    inp = sc.assignment_vector({"X": 77, "Y": 1234567, "Z": _mixed_expected(77, 1234567)})
    _, a, b, c = sc.run_vprogram(inp)
    pub = inp[:sc.n_public - 1]
    blind = [rng.randrange(R) for _ in range(9)]
    assert P.prove_fast(key, a, b, c, pub, blind) == P.prove(key, a, b, c, pub, blind)
