"""gnark-crypto ``fr.Hash`` / ``hash_to_field`` for BN254's scalar field [UPSTREAM-RECALL]:
RFC 9380 hash_to_field with expand_message_xmd over SHA-256, L = 16 + 32 = 48 bytes per element,
big-endian, reduced mod r.  Groth16's commitment extension uses it twice (gnark
backend/groth16/bn254/prove.go, verify.go): the value of a commitment wire is
``Hash(commitment.Marshal() || hashed public values, dst = "bsb22-commitment")`` and the folding
challenge of the proofs of knowledge is ``Hash(commitment wire values, dst = "G16-BSB22")``.
Host side of the prover (csrc/commit.hip holds the C++ twin) and of the verifier."""
import hashlib

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
COMMITMENT_DST = b"bsb22-commitment"
POK_DST = b"G16-BSB22"


def expand_message_xmd(msg: bytes, dst: bytes, length: int) -> bytes:
    b_in_bytes, s_in_bytes = 32, 64
    ell = -(-length // b_in_bytes)
    if ell > 255 or len(dst) > 255:
        raise ValueError("expand_message_xmd: output or DST too long")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(s_in_bytes) + msg + length.to_bytes(2, "big") + b"\x00" +
                        dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:length]


def hash_fr(msg: bytes, dst: bytes, count: int = 1):
    """fr.Hash(msg, dst, count) -> list of integers < r."""
    L = 48
    u = expand_message_xmd(msg, dst, count * L)
    return [int.from_bytes(u[i * L:(i + 1) * L], "big") % R for i in range(count)]


def g1_marshal(pt) -> bytes:
    """G1Affine.Marshal() = RawBytes(): X || Y big-endian canonical; the point at infinity is the
    flag byte 0x40 followed by zeros (gnark-crypto marshal.go mUncompressedInfinity, as recalled)."""
    if pt is None:
        return b"\x40" + bytes(63)
    return int(pt[0]).to_bytes(32, "big") + int(pt[1]).to_bytes(32, "big")


def commitment_challenge(commitment_pt, hashed_values) -> int:
    """value of a commitment wire: hash of the Pedersen commitment and the committed public
    values (constraint.SerializeCommitment + HashToFieldFn of gnark's prover / verifier)."""
    msg = g1_marshal(commitment_pt) + b"".join(int(v).to_bytes(32, "big") for v in hashed_values)
    return hash_fr(msg, COMMITMENT_DST, 1)[0]


def pok_challenge(commitment_wire_values) -> int:
    """folding challenge of pedersen.BatchProve / FoldCommitments"""
    msg = b"".join(int(v).to_bytes(32, "big") for v in commitment_wire_values)
    return hash_fr(msg, POK_DST, 1)[0]
