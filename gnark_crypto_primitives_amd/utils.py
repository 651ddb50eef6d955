"""Hasher adaptors (reference utils/hashers.go:10-27)."""
from .hash import poseidon as _poseidon


def PoseidonHasher(api, *data):
    """utils.PoseidonHasher (utils/hashers.go:25-27)."""
    return _poseidon.Hash(api, *data)


def PoseidonMultiHasher(api, *data):
    """``poseidon.MultiHash`` as a utils.Hasher (the reference's elgamal tests use
    ``HashFn = poseidon.MultiHash``, elgamal/ciphertext_test.go:272)."""
    return _poseidon.MultiHash(api, *data)
