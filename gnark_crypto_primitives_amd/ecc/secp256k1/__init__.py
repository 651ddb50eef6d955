"""secp256k1 gadgets (reference ecc/secp256k1/ecdsa)."""
from .ecdsa import DeriveAddress, PublicKey  # noqa: F401
