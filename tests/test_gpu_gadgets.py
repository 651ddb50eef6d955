"""GPU (HIP path, through the C-ABI) vs C-oracle proofs for the gadget rows that only had CPU tests
in round 1: smt.Processor (tree/smt/processor.go:10-72),
DecryptionProof.Verify on the reference's hard-coded vector K3 and AssertDecrypt
(elgamal/ciphertext.go:50-67, 124-168), eddsa.Verifier (ecc/bn254/eddsa/verifier.go:55-88) and
poseidon.MultiHash with 17 inputs (hash/native/bn254/poseidon/poseidon.go:54-91).

Every test solves + proves the batch on the GPU and compares status and every proof bit for bit
with the oracle's constraint-by-constraint solver + Groth16 prover (tests/test_gpu_prove.py::
_prove_and_check); invalid assignments of the reference's own negative tests (K5) are in the
batches and must come back ZKMI_ERR_UNSATISFIED.
"""
import json
import os
import random

import pytest

from gnark_crypto_primitives_amd import circuits
from gnark_crypto_primitives_amd.ecc import babyjub_native as bjj
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.tree.smt_wrapper import MemTree, WrapperArbo, delete_assignment
from tests import helpers as H
from tests.test_gpu_prove import _prove_and_check

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_processor_insert_update_delete_k5(zk_ctx):
    """Processor assignments from the WrapperArbo restatement on a 12-level tree: inserts into
    empty slots and next to existing leaves, updates, circomlib deletes, the all-zero NOP of
    processor_test.go:47-58 (valid) and IsOld0 = 2 of :60-70 (invalid), a wrong new root."""
    from tests.test_elgamal_processor import _processor_circuit
    from tests.test_smt_wrapper import processor_inputs
    levels = 12
    cc = compile_circuit(_processor_circuit(levels))
    rng = random.Random(120)
    w = WrapperArbo(MemTree(levels), levels)
    asg, keys, kinds = [], [], set()
    for step in range(60):
        k = rng.choice(keys) if keys and step % 3 == 2 else rng.randrange(1 << levels)
        a = w.Set(k, rng.randrange(1, 1 << 64))
        keys.append(k)
        kinds.add((a.Fnc0, a.Fnc1, a.IsOld0))
        asg.append(processor_inputs(a))
        if (a.Fnc0, a.Fnc1) == (1, 0) and step % 2:
            asg.append(processor_inputs(delete_assignment(a)))
    assert kinds >= {(1, 0, 1), (1, 0, 0), (0, 1, 0)}
    zero = dict(OldRoot=0, Siblings=[0] * levels, OldKey=0, OldValue=0, IsOld0=0, NewKey=0,
                NewValue=0, Fnc0=0, Fnc1=0, NewRoot=0)
    n0 = len(asg)
    asg += [zero, dict(zero, IsOld0=2), dict(asg[5], NewRoot=(asg[5]["NewRoot"] + 1) % H.R)]
    status = _prove_and_check(zk_ctx, cc, asg, 61)
    assert list(status != 0) == [i in (n0 + 1, n0 + 2) for i in range(len(asg))]


def test_decryption_proof_k3_and_invalid_a1y(zk_ctx):
    """elgamal/ciphertext_test.go:289-344 on the GPU: the hard-coded Chaum-Pedersen proof (K3)
    proves; A1.Y = 0 is unsatisfied; fresh proofs for random keys prove."""
    from gnark_crypto_primitives_amd.hash import poseidon_native
    v = {k: int(x) for k, x in json.load(open(os.path.join(GOLD, "chaum_pedersen_k3.json"))).items()}
    cc = compile_circuit(circuits.DecryptionProofCircuit())
    k3 = {"PubKey": [v["pubKeyX"], v["pubKeyY"]],
          "Ct": [v["c1X"], v["c1Y"], v["c2X"], v["c2Y"]],
          "A1": [v["mockA1X"], v["mockA1Y"]], "A2": [v["mockA2X"], v["mockA2Y"]],
          "Z": v["mockZ"], "Msg": v["mockMsg"]}
    asg = [k3, dict(k3, A1=[v["mockA1X"], 0])]
    rng = random.Random(33)
    for _ in range(3):      # a prover's own Chaum-Pedersen proof of correct decryption
        sk = rng.randrange(1, bjj.ORDER)
        pub = bjj.mul(bjj.BASE, sk)
        k, msg = rng.randrange(1, bjj.ORDER), rng.getrandbits(32)
        c1 = bjj.mul(bjj.BASE, k)
        c2 = bjj.add(bjj.mul(bjj.BASE, msg), bjj.mul(pub, k))
        d = bjj.mul(c1, sk)
        r = rng.randrange(1, bjj.ORDER)
        a1, a2 = bjj.mul(bjj.BASE, r), bjj.mul(c1, r)
        e = poseidon_native.multihash([*pub, *pub, *c1, *d, *a1, *a2])
        z = (r + e * sk) % bjj.ORDER
        asg.append({"PubKey": list(pub), "Ct": list(c1 + c2), "A1": list(a1), "A2": list(a2),
                    "Z": z, "Msg": msg})
    asg.append(dict(asg[2], Msg=asg[2]["Msg"] + 1))
    status = _prove_and_check(zk_ctx, cc, asg, 62)
    assert list(status != 0) == [False, True, False, False, False, True]


def test_encrypt_assert_decrypt(zk_ctx):
    """TestEncryptAssertDecrypt (elgamal/ciphertext_test.go:205-284): Encrypt + AssertDecrypt in
    one circuit; wrong private key / message / k are unsatisfied."""
    from tests.test_elgamal_processor import _ct
    from gnark_crypto_primitives_amd.elgamal import Ciphertext
    from gnark_crypto_primitives_amd.frontend import Public, Secret
    from gnark_crypto_primitives_amd.std.twistededwards import Point

    class Circuit:
        PubKey = Public(2)
        Result = Public(4)
        PrivKey = Secret()
        K = Secret()
        Msg = Secret()

        def define(self, api):
            res = Ciphertext().Encrypt(api, Point(*self.PubKey), self.K, self.Msg)
            res.AssertIsEqual(api, _ct(self.Result))
            res.AssertDecrypt(api, self.PrivKey, self.Msg)
    cc = compile_circuit(Circuit())
    rng = random.Random(31)
    asg = []
    for _ in range(3):
        priv = rng.randrange(1, bjj.ORDER)
        pub = bjj.mul(bjj.BASE, priv)
        k, msg = rng.getrandbits(160) % bjj.ORDER, rng.getrandbits(20)
        ct = bjj.mul(bjj.BASE, k) + bjj.add(bjj.mul(bjj.BASE, msg), bjj.mul(pub, k))
        asg.append({"PubKey": list(pub), "Result": list(ct), "PrivKey": priv, "K": k, "Msg": msg})
    g = asg[0]
    asg += [dict(g, PrivKey=g["PrivKey"] + 1), dict(g, Msg=g["Msg"] + 1), dict(g, K=g["K"] + 1)]
    status = _prove_and_check(zk_ctx, cc, asg, 63)
    assert list(status != 0) == [False] * 3 + [True] * 3


def test_eddsa_verifier(zk_ctx):
    """eddsa.Verifier with the Poseidon hasher against off-circuit iden3-style signatures; a
    changed message and a changed S are unsatisfied."""
    from gnark_crypto_primitives_amd.ecc import eddsa
    from gnark_crypto_primitives_amd.hash import poseidon_native
    cc = compile_circuit(circuits.EdDSACircuit())
    rng = random.Random(6)
    asg = []
    for _ in range(4):
        sk, nonce, msg = rng.randrange(bjj.ORDER), rng.randrange(bjj.ORDER), rng.randrange(H.R)
        a, r8, S = eddsa.sign_native(sk, nonce, msg, poseidon_native.hash)
        asg.append({"A": list(a), "R": list(r8), "S": S, "Msg": msg})
    asg += [dict(asg[0], Msg=(asg[0]["Msg"] + 1) % H.R), dict(asg[1], S=(asg[1]["S"] + 1) % bjj.ORDER)]
    status = _prove_and_check(zk_ctx, cc, asg, 64)
    assert list(status != 0) == [False] * 4 + [True] * 2


def test_multihash_17_and_40_inputs(zk_ctx):
    """poseidon.MultiHash beyond one permutation: 17 inputs (16 + 1, then a 2-input hash) and 40
    inputs (16 + 16 + 8), expected digests from the Python oracle."""
    from oracle import pyref
    from tests.test_poseidon import _hash_circuit
    rng = random.Random(17)
    for n in (17, 40):
        cc = compile_circuit(_hash_circuit(n, multi=True))
        asg = []
        for i in range(5):
            ins = list(range(5, 5 + n)) if i == 0 else [rng.randrange(H.R) for _ in range(n)]
            asg.append({"In": ins, "Out": pyref.poseidon_multihash(ins)})
        asg.append(dict(asg[1], Out=(asg[1]["Out"] + 1) % H.R))
        status = _prove_and_check(zk_ctx, cc, asg, 65 + n)
        assert list(status != 0) == [False] * 5 + [True]
