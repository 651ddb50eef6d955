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
