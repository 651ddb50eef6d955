"""Off-circuit iden3 MiMC7 over BN254's scalar field (x^7, 91 rounds, Miyaguchi-Preneel sponge) and
its round constants.

The reference's gadgets (hash/native/bn254/mimc7/mimc.go:22-87, hash/emulated/bn254/mimc7/mimc.go)
take their constants from a 90-entry decimal table (constants.go:26-116) that was "generated [by]
the method generateConstatsData of iden3 mimc7" (constants.go:23-25): c_0 = 0, then iterated
Keccak-256 from the seed "mimc", each digest reduced mod r.  ``constants()`` regenerates them;
tests/test_mimc7.py asserts equality with the reference's table as text.  The reference's tests
compute expected values with iden3's mimc7.Hash (mimc_test.go:38,97): ``hash`` is that function.
"""
import functools

from ..ecc.secp256k1.native import keccak256

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
N_ROUNDS = 91
SEED = b"mimc"


@functools.lru_cache(maxsize=None)
def constants():
    c = keccak256(SEED)
    out = [0]
    for _ in range(1, N_ROUNDS):
        c = keccak256(c)
        out.append(int.from_bytes(c, "big") % R)
    return tuple(out)


def encrypt(m, key):
    """MIMC7HashGeneric: 91 rounds of x <- (x + key + c_i)^7, then + key"""
    x = m % R
    for c in constants():
        x = pow((x + key + c) % R, 7, R)
    return (x + key) % R


def hash(inputs, key=0):
    """iden3 mimc7.Hash(arr, key): h <- h + x + E_h(x) over the inputs"""
    h = key % R
    for x in inputs:
        h = (h + x + encrypt(x, h)) % R
    return h
