"""Hash gadgets (reference: hash/hash.go:9-18 interface, hash/native/hashes.go:12 constructor)."""
from .poseidon import Poseidon, Hash as PoseidonHash, MultiHash as PoseidonMultiHash  # noqa: F401
