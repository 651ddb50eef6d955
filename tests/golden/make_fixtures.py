#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.  Run in the build container (needs /root/reference).

* chaum_pedersen_k3.json -- the 13 integers hard-coded in the reference's
  elgamal/ciphertext_test.go:289-303, extracted as data by regex.
* poseidon_kat.json -- (inputs, expected) pairs: the public circomlib/iden3 vectors K1/K2 with the
  inputs the reference's tests use (hash/native/bn254/poseidon/poseidon_test.go:39,
  hash/emulated/bn254/poseidon/poseidon_test.go:52-54,86) and expected values computed by the
  textbook Poseidon of oracle/pyref.py (itself pinned by the public K1 vector).
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402

REF = "/root/reference"


def chaum_pedersen():
    src = open(os.path.join(REF, "elgamal/ciphertext_test.go")).read()
    body = src[src.index("func TestVerifyDecryptionProof"):]
    out = {}
    for name, val in re.findall(r'(\w+), _ := new\(big\.Int\)\.SetString\("(\d+)", 10\)', body):
        out[name] = val
    assert len(out) == 12 and "mockMsg" in out, out.keys()
    return out


def poseidon():
    cases = [[1, 2], [1], [1, 2, 3], [297262668938251460872476410954775437897592223497],
             list(range(1, 17)), list(range(1, 61))]
    return [{"inputs": [str(x) for x in c], "hash": str(pyref.poseidon_multihash(c))}
            for c in cases]


if __name__ == "__main__":
    json.dump(chaum_pedersen(), open(os.path.join(HERE, "chaum_pedersen_k3.json"), "w"), indent=1)
    json.dump(poseidon(), open(os.path.join(HERE, "poseidon_kat.json"), "w"), indent=1)
    print("fixtures written")


def groth16_regression():
    """Self-regression fixture: seeded setup + (r, s) -> proof bytes, produced
    by the C oracle for the single-Poseidon circuit (BASELINE config 1) and a 6-level SMT
    inclusion circuit.  Guards both the oracle and the GPU path against silent drift."""
    import random

    import numpy as np

    from gnark_crypto_primitives_amd import circuits, groth16
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
    from gnark_crypto_primitives_amd.tree import smt_witness
    from oracle import cref
    from tests import helpers as H
    out = {}
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    for name, circuit, seed in (("poseidon", circuits.PoseidonCircuit(), 101),
                                ("smt6", circuits.smt_inclusion_circuit(6), 102)):
        cc = compile_circuit(circuit)
        pk, _, _ = groth16.setup(cc, seed, mul)
        rng = random.Random(seed)
        if name == "poseidon":
            datas = [rng.randrange(pyref.R) for _ in range(3)]
            asg = [{"Data": d, "Hash": pyref.poseidon_hash([d])} for d in datas]
        else:
            asg = [smt_witness.synthetic_inclusion(rng, 6, k) for k in (0, 2, 5)]
        inputs = [cc.assignment_vector(a) for a in asg]
        rs = [[rng.randrange(pyref.R), rng.randrange(pyref.R)] for _ in asg]
        proofs, status, _ = cref.groth16_prove_batch(
            cref.R1csHandle(cc), cref.PkHandle(pk),
            np.stack([to_mont_array(v) for v in inputs]), np.stack([to_mont_array(v) for v in rs]))
        assert not status.any()
        out[name] = {"setup_seed": seed, "fingerprint": cc.fingerprint(),
                     "inputs": [[str(x) for x in v] for v in inputs],
                     "rs": [[str(x) for x in v] for v in rs],
                     "proofs_hex": [p.tobytes().hex() for p in proofs]}
    return out


if __name__ == "__main__":
    json.dump(groth16_regression(), open(os.path.join(HERE, "groth16_regression.json"), "w"),
              indent=1)
    print("groth16 regression fixture written")
