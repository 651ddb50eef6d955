"""Poseidon: parameter tables, known-answer vectors, gadget vs off-circuit vs textbook oracle."""
import json
import os
import random
import re

import pytest

from gnark_crypto_primitives_amd.frontend import Public, Secret, compile_circuit
from gnark_crypto_primitives_amd.hash import poseidon, poseidon_native
from gnark_crypto_primitives_amd.hash.poseidon_constants import N_ROUNDS_P, opt_params
from oracle import pyref

GOLD = os.path.join(os.path.dirname(__file__), "golden")
REF_CONSTANTS = "/root/reference/hash/native/bn254/poseidon/constants.go"

# public circomlib / iden3 vectors
K1 = 7853200120776062878684798364095072458815029376092732009249414926327459813530
K2A = 18586133768512220936620570745912940619677854269274689475585506675881198879027
K2B = 6542985608222806190361240322586112750744169038454362455181422643027100751666


def test_oracle_pinned_by_public_kats():
    assert pyref.poseidon_hash([1, 2]) == K1
    assert pyref.poseidon_hash([1]) == K2A
    assert pyref.poseidon_hash([1, 2, 3]) == K2B


def test_native_matches_golden_and_oracle():
    for case in json.load(open(os.path.join(GOLD, "poseidon_kat.json"))):
        ins = [int(x) for x in case["inputs"]]
        assert poseidon_native.multihash(ins) == int(case["hash"])
        assert pyref.poseidon_multihash(ins) == int(case["hash"])
    rng = random.Random(7)
    for t in range(1, 17):
        ins = [rng.randrange(pyref.R) for _ in range(t)]
        assert poseidon_native.hash(ins) == pyref.poseidon_hash(ins)


@pytest.mark.skipif(not os.path.exists(REF_CONSTANTS), reason="reference tree not present")
def test_generated_tables_equal_reference_text():
    """Grain LFSR + optimised-constant derivation reproduce constants.go:48 (C), :2291 (M),
    :4414 (S), :22733 (P) for every width t = 2..17."""
    src = open(REF_CONSTANTS).read()

    def section(name):
        i = src.index("var " + name + " =")
        j = src.find("\nvar ", i + 5)
        return [int(x) for x in re.findall(r'"(\d+)"', src[i:j if j > 0 else len(src)])]
    allc, allm, alls, allp = (section(n) for n in ("strC", "strM", "strS", "strP"))
    oc = om = os_ = op = 0
    for t in range(2, 18):
        rp = N_ROUNDS_P[t - 2]
        c, m, p, s = opt_params(t)
        assert allc[oc:oc + 8 * t + rp] == c
        assert allm[om:om + t * t] == [x for r in m for x in r]
        assert allp[op:op + t * t] == [x for r in p for x in r]
        assert alls[os_:os_ + (2 * t - 1) * rp] == s
        oc, om, op, os_ = oc + 8 * t + rp, om + t * t, op + t * t, os_ + (2 * t - 1) * rp
    assert (oc, om, op, os_) == (len(allc), len(allm), len(allp), len(alls))


def _hash_circuit(n, multi=False):
    class Cc:
        In = Secret(n)
        Out = Public()

        def define(self, api):
            fn = poseidon.MultiHash if multi else poseidon.Hash
            api.AssertIsEqual(fn(api, *self.In), self.Out)
    return Cc()


@pytest.mark.parametrize("n,count", [(1, 214), (2, 241), (3, 262)])
def test_gadget_constraint_counts_and_kats(n, count):
    """1 input: 213 + 1 assertion; Hash2: 240 + 1; three variable inputs: 261 + 1 (Hash1(k, v, 1) folds the S-box of its constant
    input: 258)."""
    cc = compile_circuit(_hash_circuit(n))
    assert cc.n_constraints == count
    ins = [1, 2, 3][:n]
    want = {1: K2A, 2: K1, 3: K2B}[n]
    wires, a, b, c = cc.run_program([want] + ins)
    assert cc.is_satisfied(wires)[0] and cc.last_status == 0
    wires, *_ = cc.run_program([want + 1] + ins)
    assert cc.last_status != 0


def test_gadget_multihash_17_inputs():
    ins = list(range(5, 22))
    cc = compile_circuit(_hash_circuit(17, multi=True))
    wires, *_ = cc.run_program([pyref.poseidon_multihash(ins)] + ins)
    assert cc.is_satisfied(wires)[0] and cc.last_status == 0


def test_write_drops_overflow_and_empty_hash_errors():
    """poseidon.go:103-108 (Write silently ignores a write that would exceed 16 inputs) and
    :41-43 (Hash of nothing is an error)."""
    from gnark_crypto_primitives_amd.frontend.api import API
    api = API()
    h = poseidon.Poseidon(api)
    h.Write(*range(1, 11))
    h.Write(*range(1, 11))
    assert len(h.data) == 10
    with pytest.raises(ValueError):
        poseidon.Hash(api)
    with pytest.raises(ValueError):
        poseidon.MultiHash(api, *range(4097))


def test_k3_chaum_pedersen_vector():
    """The only fully hard-coded vector of the reference (elgamal/ciphertext_test.go:289-303):
    pins Poseidon t = 13, the BabyJubJub parameters and the (P, P, C1, D, A1, A2) hashing order of
    elgamal/ciphertext.go:146."""
    v = {k: int(x) for k, x in json.load(open(os.path.join(GOLD, "chaum_pedersen_k3.json"))).items()}
    P = (v["pubKeyX"], v["pubKeyY"])
    C1, C2 = (v["c1X"], v["c1Y"]), (v["c2X"], v["c2Y"])
    A1, A2 = (v["mockA1X"], v["mockA1Y"]), (v["mockA2X"], v["mockA2Y"])
    for pt in (P, C1, C2, A1, A2, pyref.BJJ_BASE):
        assert pyref.bjj_on_curve(pt)
    M = pyref.bjj_mul(pyref.BJJ_BASE, v["mockMsg"])
    D = pyref.bjj_add(C2, (-M[0] % pyref.R, M[1]))
    e = pyref.poseidon_multihash([*P, *P, *C1, *D, *A1, *A2])
    assert e == 10507737167015891178547203577303854462031834800541656436509630665109174295651
    assert poseidon_native.multihash([*P, *P, *C1, *D, *A1, *A2]) == e
    z = v["mockZ"]
    assert pyref.bjj_mul(pyref.BJJ_BASE, z) == pyref.bjj_add(A1, pyref.bjj_mul(P, e))
    assert pyref.bjj_mul(C1, z) == pyref.bjj_add(A2, pyref.bjj_mul(D, e))
