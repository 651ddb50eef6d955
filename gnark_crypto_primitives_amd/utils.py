"""Hasher adaptors (reference utils/hashers.go:10-27)."""
from .hash import poseidon as _poseidon


def PoseidonHasher(api, *data):
    """utils.PoseidonHasher (utils/hashers.go:25-27)."""
    return _poseidon.Hash(api, *data)
