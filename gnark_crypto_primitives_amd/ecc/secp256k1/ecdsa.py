"""``ecdsa.DeriveAddress`` (reference ecc/secp256k1/ecdsa/address.go:14-40): Ethereum address of a
secp256k1 public key inside a circuit.  X and Y arrive as emulated elements (4 x 64-bit limbs,
least significant first); their bytes are put in big-endian order, hashed with legacy Keccak-256,
and bytes 12..31 of the digest are folded big-endian into one variable."""
from ... import utils
from ...std import sha3


class PublicKey:
    """gnark std/signature/ecdsa.PublicKey[Secp256k1Fp, Secp256k1Fr]: an affine point, X and Y
    emulated.Element."""
    def __init__(self, X, Y):
        self.X, self.Y = X, Y


def DeriveAddress(api, pub_key, commit=False, byte_tables=False):
    """commit: byte range checks through gnark's commitment-based checker (what ``uints.New`` gives
    an R1CS builder) instead of boolean wires.  byte_tables: the whole gadget the way gnark compiles
    it -- Keccak over uints.U64 with the XOR / AND lookup tables (std/sha3.py::permute_bytes) instead
    of this repo's boolean Keccak (implies commit)."""
    commit = commit or byte_tables
    x_bytes = utils.ElemToU8(api, pub_key.X, commit)
    y_bytes = utils.ElemToU8(api, pub_key.Y, commit)
    pub_bytes = utils.SwapEndianness(x_bytes) + utils.SwapEndianness(y_bytes)
    keccak = sha3.NewLegacyKeccak256Bytes(api) if byte_tables else sha3.NewLegacyKeccak256(api)
    keccak.Write(pub_bytes)
    digest = keccak.Sum()
    return utils.U8ToVar(api, digest[12:])
