"""SMT verifier gadget: satisfiable on valid paths, unsatisfiable on the reference's negative
cases; both solvers agree."""
import random

import pytest

from gnark_crypto_primitives_amd import circuits
from gnark_crypto_primitives_amd.frontend import Public, Secret, compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import from_mont_array, to_mont_array
from gnark_crypto_primitives_amd.tree import smt, smt_witness
from oracle import cref, pyref


@pytest.fixture(scope="module")
def incl12():
    return compile_circuit(circuits.smt_inclusion_circuit(12))


def test_size_matches_survey_estimate():
    """~244 constraints per level + two leaf hashes (n=64 -> ~16.3 k)."""
    cc = compile_circuit(circuits.smt_inclusion_circuit(64))
    assert 16000 < cc.n_constraints < 16700
    assert cc.domain_log2() == 15


@pytest.mark.parametrize("populated", [0, 1, 5, 11])
def test_inclusion_valid_and_oracle_agreement(incl12, populated):
    rng = random.Random(populated)
    w = smt_witness.synthetic_inclusion(rng, 12, populated)
    assert w["Root"] == pyref.smt_root_from_path(w["Key"], w["Value"], w["Siblings"])
    inp = incl12.assignment_vector(w)
    wires, a, b, c = incl12.run_program(inp)
    assert incl12.last_status == 0 and incl12.is_satisfied(wires)[0]
    rc, w2, a2, b2, c2 = cref.r1cs_solve(cref.R1csHandle(incl12), to_mont_array(inp))
    assert rc == 0 and from_mont_array(w2) == wires and from_mont_array(c2) == c


def test_inclusion_invalid_cases(incl12):
    rng = random.Random(9)
    rh = cref.R1csHandle(incl12)
    good = smt_witness.synthetic_inclusion(rng, 12, 4)

    def bad(mut):
        w = dict(good, Siblings=list(good["Siblings"]))
        mut(w)
        inp = incl12.assignment_vector(w)
        incl12.run_program(inp)
        return incl12.last_status != 0 and cref.r1cs_solve(rh, to_mont_array(inp))[0] < 0
    assert bad(lambda w: w.update(Root=(w["Root"] + 1) % pyref.R))
    assert bad(lambda w: w.update(Value=w["Value"] ^ 1))
    assert bad(lambda w: w.update(Key=w["Key"] ^ 1))            # flips the level-0 direction
    assert bad(lambda w: w["Siblings"].__setitem__(11, 5))      # last sibling must be 0
    assert bad(lambda w: w["Siblings"].__setitem__(2, 0))       # hole in the path changes the root
    assert bad(lambda w: w.update(Key=w["Key"] + (1 << 12)))    # key wider than the tree


def test_lowbits_is_constrained():
    """tree/smt/utils_test.go:30-39: key 5 must not decompose to bits (1, 1, 1)."""
    class Cc:
        Key = Secret()
        B = Secret(3)

        def define(self, api):
            for got, want in zip(smt.lowBits(api, self.Key, 3), self.B):
                api.AssertIsEqual(got, want)
    cc = compile_circuit(Cc())
    cc.run_program(cc.assignment_vector({"Key": 5, "B": [1, 0, 1]}))
    assert cc.last_status == 0
    cc.run_program(cc.assignment_vector({"Key": 5, "B": [1, 1, 1]}))
    assert cc.last_status != 0


def test_verifier_inclusion_and_exclusion_share_one_system():
    """Config 3: fnc = 0 and fnc = 1 through smt.Verifier (tree/smt/verifier.go:66-81,102)."""
    levels = 10
    cc = compile_circuit(circuits.smt_verifier_circuit(levels))
    rng = random.Random(4)
    w = smt_witness.synthetic_inclusion(rng, levels, 3)
    incl = dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Fnc=0)
    cc.run_program(cc.assignment_vector(incl))
    assert cc.last_status == 0
    # exclusion of key' that shares the first 3 path bits with an existing leaf (old key/value)
    other = (w["Key"] & 0b111) | (((w["Key"] >> 3) ^ 1) << 3)
    excl = dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Key=other, Value=0, Fnc=1)
    cc.run_program(cc.assignment_vector(excl))
    assert cc.last_status == 0
    # claiming exclusion of the very key that is present must fail (key-reuse guard)
    excl_same = dict(excl, Key=w["Key"])
    cc.run_program(cc.assignment_vector(excl_same))
    assert cc.last_status != 0
    # exclusion against an empty branch: isOld0 = 1, root = fold with the zero leaf
    sib = [rng.randrange(1, pyref.R) for _ in range(2)] + [0] * (levels - 2)
    key = rng.getrandbits(levels)
    cur = 0
    for i in (1, 0):
        cur = pyref.poseidon_hash([sib[i], cur]) if (key >> i) & 1 else \
            pyref.poseidon_hash([cur, sib[i]])
    empty = dict(Root=cur, OldKey=0, OldValue=0, IsOld0=1, Key=key, Value=0, Fnc=1, Siblings=sib)
    cc.run_program(cc.assignment_vector(empty))
    assert cc.last_status == 0
