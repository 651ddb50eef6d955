"""iden3 MiMC7 (reference hash/native/bn254/mimc7, hash/emulated/bn254/mimc7): constants regenerated
from the Keccak chain equal the reference's table as text; public iden3 vectors ([UPSTREAM-RECALL]
of go-iden3-crypto's mimc7_test.go) pin the off-circuit function the reference's tests use for
expected values (mimc_test.go:38,97); the gadgets reproduce it -- the reference's own test circuits
(mimc_test.go:18-31: preimage 12; :52-113: 62 inputs, and 63 inputs silently dropped)."""
import os
import random
import re

import pytest

from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import Public, Secret
from gnark_crypto_primitives_amd.hash import emulated_mimc7, mimc7, mimc7_native
from gnark_crypto_primitives_amd.std import emulated as em

REF_CONSTANTS = "/root/reference/hash/native/bn254/mimc7/constants.go"
REF_CONSTANTS_EMULATED = "/root/reference/hash/emulated/bn254/mimc7/constants.go"


@pytest.mark.skipif(not os.path.exists(REF_CONSTANTS), reason="reference tree not present")
def test_generated_constants_equal_reference_text():
    for path in (REF_CONSTANTS, REF_CONSTANTS_EMULATED):
        ref = [int(x) for x in re.findall(r'"(\d{20,})"', open(path).read())]
        ref = [x for x in ref if x != mimc7_native.R]       # the emulated file also holds q
        assert len(ref) == 90 and tuple([0] + ref) == mimc7_native.constants()


def test_iden3_vectors():
    assert mimc7_native.constants()[1] == \
        20888961410941983456478427210666206549300505294776164667214940546594746570981
    assert mimc7_native.encrypt(1, 2) == \
        10594780656576967754230020536574539122676596303354946869887184401991294982664
    assert mimc7_native.hash([12]) == \
        0x237c92644dbddb86d8a259e0e923aaab65a93f1ec5758b8799988894ac0958fd
    assert mimc7_native.hash([78, 41]) == \
        0x067f3202335ea256ae6e6aadcd2d5f7f4b06a00b2d1e0de903980d5ab552dc70
    assert mimc7_native.hash([12, 45, 78, 41]) == \
        0x284bc1f34f335933a23a433b6ff3ee179d682cd5e5e2fcdd2d964afa85104beb


class MiMCCircuit:
    """testMiMCCircuit (mimc_test.go:18-31) / testMaxInputsMiMCCircuit (:52-65)"""
    Hash = Public()

    def __init__(self, n):
        self.n = n
        type(self).Preimages = Secret(n)

    def define(self, api):
        h = mimc7.New(api)
        h.Write(*self.Preimages)
        if self.n <= mimc7.MAX_INPUTS:
            h.AssertSumIsEqual(self.Hash)
        else:                                   # testLimitInputsMiMCCircuit (:67-80): the write is dropped
            assert not h.WriteSucceeded()
            api.AssertIsEqual(self.Hash, self.Hash)


def _circuit(n):
    return type("MiMCCircuit%d" % n, (MiMCCircuit,), {})(n)


def test_native_gadget_reference_circuits():
    rng = random.Random(7)
    cc = compile_circuit(_circuit(1))
    assert cc.n_constraints == 4 * 91 + 3          # 4 per round, IsZero (2), flag == 1
    w, *_ = cc.run_vprogram(cc.assignment_vector({"Hash": mimc7_native.hash([12]), "Preimages": [12]}))
    assert cc.last_status == 0 and cc.is_satisfied(w)[0]
    cc.run_vprogram(cc.assignment_vector({"Hash": mimc7_native.hash([13]), "Preimages": [12]}))
    assert cc.last_status != 0
    x = rng.randrange(mimc7_native.R)
    cc = compile_circuit(_circuit(62))
    w, *_ = cc.run_vprogram(cc.assignment_vector({"Hash": mimc7_native.hash([x] * 62),
                                                  "Preimages": [x] * 62}))
    assert cc.last_status == 0 and cc.is_satisfied(w)[0]
    cc = compile_circuit(_circuit(63))             # compiles; nothing was hashed
    assert cc.n_constraints <= 1


class EmulatedMiMCCircuit:
    """hash/emulated/bn254/mimc7/mimc_test.go:19-37"""
    Hash = Public(4)
    Preimage = Secret(4)

    def define(self, api):
        sf = emulated_mimc7.ScalarField
        h = emulated_mimc7.New(api)
        h.Write(em.Element(self.Preimage, sf))
        h.AssertSumIsEqual(em.Element(self.Hash, sf))

    @staticmethod
    def assignment(x, h=None):
        sf = emulated_mimc7.ScalarField
        return {"Hash": em.ValueOf(mimc7_native.hash([x]) if h is None else h, sf),
                "Preimage": em.ValueOf(x, sf)}


def test_emulated_gadget_matches_native():
    cc = compile_circuit(EmulatedMiMCCircuit(), 16)
    w, *_ = cc.run_vprogram(cc.assignment_vector(EmulatedMiMCCircuit.assignment(12)))
    assert cc.last_status == 0 and cc.is_satisfied(w)[0]
    cc.run_vprogram(cc.assignment_vector(EmulatedMiMCCircuit.assignment(12, mimc7_native.hash([13]))))
    assert cc.last_status != 0
