"""BabyJubJub ElGamal gadgets (reference package elgamal/)."""
from .ciphertext import Ciphertext, DecryptionProof, NewCiphertext, hashPointsToScalar  # noqa: F401
from .encrypt import EncryptedZero  # noqa: F401
from .mul import FixedBaseScalarMulBN254  # noqa: F401
