"""gnark / gnark-crypto wire formats for the artefacts that cross the prover boundary
(first slice): group-element encodings, the Groth16 proof, the witness vector.

Everything here restates upstream formats from memory [UPSTREAM-RECALL]: gnark-crypto's bn254
``G1Affine.Bytes/RawBytes`` and ``G2Affine.Bytes/RawBytes`` (marshal.go), gnark's groth16 bn254
``Proof.WriteTo/WriteRawTo`` and ``witness.WriteTo``.  No gnark build exists offline to confirm
them; the tests check internal consistency (round trips, flag semantics, sizes).

Encoding rules restated:
* field elements are written big-endian, canonical (non-Montgomery) form, 32 bytes;
* a compressed G1 point is X (32 B) with the two most significant bits of the first byte as flags:
  0b10 = compressed, Y is the lexicographically smallest root; 0b11 = compressed, largest root;
  0b01 = point at infinity (compressed form); 0b00 = uncompressed (X || Y, 64 B; infinity = 64 zero
  bytes, and the reader also accepts the 0b01 spelling);
* a G2 point writes X.A1 then X.A0 (then Y.A1, Y.A0 when uncompressed); "largest" compares A1
  first, then A0;
* Proof.WriteTo = Ar (compressed) | Bs (compressed) | Krs (compressed) | uint32 number of
  commitments (0 here) ... = 128 + 4 bytes (+ commitment PoK when present);
  Proof.WriteRawTo uses the uncompressed forms: 64 + 128 + 64 + 4;
* witness.WriteTo = uint32 nbPublic | uint32 nbSecret | uint32 length | length x 32-byte elements.
"""
from __future__ import annotations

import struct

import numpy as np

from .frontend.compile import array_to_ints, ints_to_array

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_MONT = (1 << 256) % P
_MONT_INV = pow(_MONT, P - 2, P)
_MONT_R = (1 << 256) % R
_MONT_R_INV = pow(_MONT_R, R - 2, R)

M_UNCOMPRESSED, M_INFINITY, M_SMALLEST, M_LARGEST = 0b00 << 6, 0b01 << 6, 0b10 << 6, 0b11 << 6


def _fq_plain(limbs4):          # Montgomery limbs -> int
    return array_to_ints(np.asarray(limbs4, dtype=np.uint64).reshape(1, 4))[0] * _MONT_INV % P


def _fq_mont(x):
    return ints_to_array([x % P * _MONT % P])[0]


def _sqrt_fp(a):
    # p = 3 mod 4
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a % P else None


def _f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2_sqrt(a):
    """square root in Fp2 = Fp[u]/(u^2+1) (complex method), or None"""
    a0, a1 = a
    if a1 == 0:
        r = _sqrt_fp(a0)
        if r is not None:
            return (r, 0)
        r = _sqrt_fp(-a0 % P)
        return (0, r) if r is not None else None
    n = _sqrt_fp((a0 * a0 + a1 * a1) % P)
    if n is None:
        return None
    for s in (n, -n % P):
        t = (a0 + s) * pow(2, P - 2, P) % P
        x0 = _sqrt_fp(t)
        if x0 is not None and x0 != 0:
            x1 = a1 * pow(2 * x0, P - 2, P) % P
            if _f2_mul((x0, x1), (x0, x1)) == (a0 % P, a1 % P):
                return (x0, x1)
    return None


_B2 = _f2_mul((3, 0), (lambda d: (9 * d % P, -d % P))(pow(82, P - 2, P)))   # 3 / (9 + u)


def _lex_largest_fp(y):
    return y > (P - 1) // 2


def _lex_largest_fp2(y):
    return _lex_largest_fp(y[1]) if y[1] != 0 else _lex_largest_fp(y[0])


# ---- G1 ------------------------------------------------------------------------------------------
def g1_to_bytes(pt_mont, compressed=True) -> bytes:
    """pt_mont: 8 uint64 (X, Y Montgomery limbs), all zero = infinity."""
    a = np.asarray(pt_mont, dtype=np.uint64).reshape(8)
    size = 32 if compressed else 64
    if not a.any():
        # raw form: all zero bytes under the "uncompressed" flag (as recalled, gnark-crypto's decoder
        # consumes only the compressed size on flag 0b01, which would misalign a raw key)
        return bytes([M_INFINITY]) + bytes(size - 1) if compressed else bytes(size)
    x, y = _fq_plain(a[:4]), _fq_plain(a[4:])
    xb = bytearray(x.to_bytes(32, "big"))
    if not compressed:
        return bytes(xb) + y.to_bytes(32, "big")
    xb[0] |= M_LARGEST if _lex_largest_fp(y) else M_SMALLEST
    return bytes(xb)


def g1_from_bytes(buf: bytes) -> np.ndarray:
    flag = buf[0] & 0xC0
    if flag == M_INFINITY:
        return np.zeros(8, dtype=np.uint64)
    x = int.from_bytes(bytes([buf[0] & 0x3F]) + buf[1:32], "big")
    if flag == M_UNCOMPRESSED:
        y = int.from_bytes(buf[32:64], "big")
        if x == 0 and y == 0:                    # raw infinity (both spellings are accepted)
            return np.zeros(8, dtype=np.uint64)
    else:
        y = _sqrt_fp((x * x * x + 3) % P)
        if y is None:
            raise ValueError("x is not on the curve")
        if _lex_largest_fp(y) != (flag == M_LARGEST):
            y = P - y
    return np.concatenate([_fq_mont(x), _fq_mont(y)])


# ---- G2 ------------------------------------------------------------------------------------------
def g2_to_bytes(pt_mont, compressed=True) -> bytes:
    a = np.asarray(pt_mont, dtype=np.uint64).reshape(16)
    size = 64 if compressed else 128
    if not a.any():
        return bytes([M_INFINITY]) + bytes(size - 1) if compressed else bytes(size)
    x0, x1, y0, y1 = (_fq_plain(a[4 * i:4 * i + 4]) for i in range(4))
    xb = bytearray(x1.to_bytes(32, "big") + x0.to_bytes(32, "big"))
    if not compressed:
        return bytes(xb) + y1.to_bytes(32, "big") + y0.to_bytes(32, "big")
    xb[0] |= M_LARGEST if _lex_largest_fp2((y0, y1)) else M_SMALLEST
    return bytes(xb)


def g2_from_bytes(buf: bytes) -> np.ndarray:
    flag = buf[0] & 0xC0
    if flag == M_INFINITY:
        return np.zeros(16, dtype=np.uint64)
    x1 = int.from_bytes(bytes([buf[0] & 0x3F]) + buf[1:32], "big")
    x0 = int.from_bytes(buf[32:64], "big")
    if flag == M_UNCOMPRESSED:
        y1, y0 = int.from_bytes(buf[64:96], "big"), int.from_bytes(buf[96:128], "big")
        if not (x0 or x1 or y0 or y1):
            return np.zeros(16, dtype=np.uint64)
    else:
        x = (x0, x1)
        rhs = _f2_mul(_f2_mul(x, x), x)
        rhs = ((rhs[0] + _B2[0]) % P, (rhs[1] + _B2[1]) % P)
        y = _f2_sqrt(rhs)
        if y is None:
            raise ValueError("x is not on the twist")
        if _lex_largest_fp2(y) != (flag == M_LARGEST):
            y = (-y[0] % P, -y[1] % P)
        y0, y1 = y
    return np.concatenate([_fq_mont(v) for v in (x0, x1, y0, y1)])


# ---- Groth16 proof ---------------------------------------------------------------------------------
def proof_to_bytes(proof32, raw=False, commitments=None, pok=None) -> bytes:
    """proof32: the 32-word record of zkmi_prove_batch (Ar | Krs | Bs).  gnark writes Ar, Bs, Krs,
    then the commitment extension: a uint32 count, the Commitments (the same point encoding as the
    rest of the proof) and CommitmentPok [UPSTREAM-RECALL].  commitments: uint64 [n, 8], pok: [8]
    (zkmi_prove_collect_ex's record split)."""
    a = np.asarray(proof32, dtype=np.uint64).reshape(32)
    ar, krs, bs = a[0:8], a[8:16], a[16:32]
    out = g1_to_bytes(ar, not raw) + g2_to_bytes(bs, not raw) + g1_to_bytes(krs, not raw)
    if commitments is None or len(commitments) == 0:
        return out + struct.pack(">I", 0)          # no commitments
    cs = np.asarray(commitments, dtype=np.uint64).reshape(-1, 8)
    out += struct.pack(">I", len(cs)) + b"".join(g1_to_bytes(c, not raw) for c in cs)
    return out + g1_to_bytes(np.asarray(pok, dtype=np.uint64).reshape(8), not raw)


def proof_from_bytes(buf: bytes, with_commitments=False):
    """-> proof record uint64 [32]; with_commitments: (record, commitments [n, 8], pok [8] | None)"""
    raw = len(buf) >= 64 + 128 + 64 and (buf[0] & 0xC0) in (M_UNCOMPRESSED, M_INFINITY) and \
        _looks_raw(buf)
    g1n, g2n = (64, 128) if raw else (32, 64)
    ar = g1_from_bytes(buf[:g1n])
    bs = g2_from_bytes(buf[g1n:g1n + g2n])
    krs = g1_from_bytes(buf[g1n + g2n:2 * g1n + g2n])
    rec = np.concatenate([ar, krs, bs])
    if not with_commitments:
        return rec
    o = 2 * g1n + g2n
    n, = struct.unpack_from(">I", buf, o)
    o += 4
    coms = np.zeros((n, 8), dtype=np.uint64)
    for i in range(n):
        coms[i] = g1_from_bytes(buf[o:o + g1n])
        o += g1n
    pok = g1_from_bytes(buf[o:o + g1n]) if n else None
    if len(buf) != o + (g1n if n else 0):
        raise ValueError("malformed proof: trailing or missing bytes")
    return rec, coms, pok


def _looks_raw(buf):
    """raw and compressed proofs are told apart by length: raw = 64 + 128 + 64 + 4 + (n + 1 if n) * 64"""
    for g1n, g2n in ((64, 128), (32, 64)):
        o = 2 * g1n + g2n
        if len(buf) < o + 4:
            continue
        n, = struct.unpack_from(">I", buf, o)
        if len(buf) == o + 4 + (n + 1 if n else 0) * g1n:
            return g1n == 64
    return len(buf) >= 64 + 128 + 64


# ---- witness -----------------------------------------------------------------------------------------
def witness_to_bytes(values, n_public: int) -> bytes:
    """values: ints, public (without the ONE wire) then secret."""
    values = [int(v) % R for v in values]
    head = struct.pack(">III", n_public, len(values) - n_public, len(values))
    return head + b"".join(v.to_bytes(32, "big") for v in values)


def witness_from_bytes(buf: bytes):
    n_pub, n_sec, n = struct.unpack(">III", buf[:12])
    if n != n_pub + n_sec or len(buf) != 12 + 32 * n:
        raise ValueError("malformed witness")
    vals = [int.from_bytes(buf[12 + 32 * i:44 + 32 * i], "big") for i in range(n)]
    if any(v >= R for v in vals):
        raise ValueError("witness element not reduced")
    return vals, n_pub


def witness_to_inputs(buf: bytes) -> np.ndarray:
    """witness bytes -> [n_inputs, 4] Montgomery limbs for Prover.prove"""
    vals, _ = witness_from_bytes(buf)
    return ints_to_array([v * _MONT_R % R for v in vals])


# ---- Groth16 keys (second slice) ----------------------------------------------------
# gnark backend/groth16/bn254/marshal.go, restated from memory [UPSTREAM-RECALL; parity unpinned:
# nothing importable offline can confirm a byte of it].  ProvingKey.WriteRawTo:
#   fft.Domain.WriteTo   uint64 cardinality | fr cardinalityInv | fr generator | fr generatorInv |
#                        fr frMultiplicativeGen | fr frMultiplicativeGenInv | uint8 withPrecompute
#   then, through gnark-crypto's Encoder (big-endian; a slice = uint32 length + elements; points
#   uncompressed under WriteRawTo):
#   G1.Alpha, G1.Beta, G1.Delta, G1.A[], G1.B[], G1.Z[], G1.K[], G2.Beta, G2.Delta, G2.B[],
#   uint64 nbWires, uint64 NbInfinityA, uint64 NbInfinityB, InfinityA[] (uint32 length + one byte per
#   bool), InfinityB[], uint32 number of commitment keys (0 for the circuits of this repository).
# VerifyingKey.WriteRawTo: G1.Alpha, G1.Beta, G2.Beta, G2.Gamma, G1.Delta, G2.Delta, G1.K[] (uint32
#   length + points), uint32 number of commitment-committed lists (0), uint32 commitment keys (0).
def _fr_plain(limbs4):
    return array_to_ints(np.asarray(limbs4, dtype=np.uint64).reshape(1, 4))[0] * _MONT_R_INV % R


def _domain_bytes(log_n: int) -> bytes:
    n = 1 << log_n
    gen = pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)
    fr = lambda v: (v % R).to_bytes(32, "big")
    return (struct.pack(">Q", n) + fr(pow(n, R - 2, R)) + fr(gen) + fr(pow(gen, R - 2, R)) +
            fr(5) + fr(pow(5, R - 2, R)) + b"\x00")


def _read_domain(buf, o):
    n, = struct.unpack_from(">Q", buf, o)
    if n == 0 or n & (n - 1):
        raise ValueError("domain cardinality is not a power of two")
    log_n = n.bit_length() - 1
    want = _domain_bytes(log_n)
    if buf[o:o + len(want) - 1] != want[:-1]:
        raise ValueError("fft.Domain constants do not match BN254's")
    return log_n, o + len(want)


def _pts(arr, enc):
    arr = np.asarray(arr, dtype=np.uint64)
    return struct.pack(">I", arr.shape[0]) + b"".join(enc(p, compressed=False) for p in arr)


def _read_pts(buf, o, size, dec, width):
    n, = struct.unpack_from(">I", buf, o)
    o += 4
    if o + n * size > len(buf):
        raise ValueError("truncated point slice")
    out = np.zeros((n, width), dtype=np.uint64)
    for i in range(n):
        out[i] = dec(buf[o + i * size:o + (i + 1) * size])
    return out, o + n * size


def proving_key_to_bytes(pk) -> bytes:
    """groth16.ProvingKey (this package's, gnark's memory image) -> ProvingKey.WriteRawTo bytes."""
    inf_a, inf_b = pk.infinity_maps()
    g1 = lambda p: g1_to_bytes(p, compressed=False)
    g2 = lambda p: g2_to_bytes(p, compressed=False)
    out = [_domain_bytes(pk.log_n), g1(pk.g1_alpha), g1(pk.g1_beta), g1(pk.g1_delta),
           _pts(pk.g1_a, g1_to_bytes), _pts(pk.g1_b, g1_to_bytes), _pts(pk.g1_z, g1_to_bytes),
           _pts(pk.g1_k, g1_to_bytes), g2(pk.g2_beta), g2(pk.g2_delta), _pts(pk.g2_b, g2_to_bytes),
           struct.pack(">QQQ", pk.n_wires, int(inf_a.sum()), int(inf_b.sum())),
           struct.pack(">I", len(inf_a)) + inf_a.tobytes(),
           struct.pack(">I", len(inf_b)) + inf_b.tobytes(), struct.pack(">I", 0)]
    return b"".join(out)


def proving_key_from_bytes(buf: bytes, n_public: int):
    """-> groth16.ProvingKey.  ``n_public`` (wires incl. ONE) is not part of gnark's file: G1.K holds
    the private wires in order, so k_wire = n_public .. n_wires - 1."""
    from .groth16 import ProvingKey
    pk = ProvingKey()
    pk.log_n, o = _read_domain(buf, 0)
    pk.g1_alpha, pk.g1_beta, pk.g1_delta = (g1_from_bytes(buf[o + 64 * i:o + 64 * i + 64])
                                            for i in range(3))
    o += 192
    pk.g1_a, o = _read_pts(buf, o, 64, g1_from_bytes, 8)
    pk.g1_b, o = _read_pts(buf, o, 64, g1_from_bytes, 8)
    pk.g1_z, o = _read_pts(buf, o, 64, g1_from_bytes, 8)
    pk.g1_k, o = _read_pts(buf, o, 64, g1_from_bytes, 8)
    pk.g2_beta, pk.g2_delta = g2_from_bytes(buf[o:o + 128]), g2_from_bytes(buf[o + 128:o + 256])
    o += 256
    pk.g2_b, o = _read_pts(buf, o, 128, g2_from_bytes, 16)
    n_wires, n_inf_a, n_inf_b = struct.unpack_from(">QQQ", buf, o)
    o += 24
    maps = []
    for want in (n_inf_a, n_inf_b):
        n, = struct.unpack_from(">I", buf, o)
        m = np.frombuffer(buf, dtype=np.uint8, count=n, offset=o + 4)
        o += 4 + n
        if n != n_wires or int(m.sum()) != want or (m > 1).any():
            raise ValueError("infinity map does not match nbWires / NbInfinity")
        maps.append(m)
    n_ck, = struct.unpack_from(">I", buf, o)
    if n_ck:
        raise ValueError("proving keys with commitment keys are not supported")
    pk.n_wires = int(n_wires)
    pk.a_wire = np.nonzero(maps[0] == 0)[0].astype(np.uint32)
    pk.b_wire = np.nonzero(maps[1] == 0)[0].astype(np.uint32)
    pk.k_wire = np.arange(n_public, n_wires, dtype=np.uint32)
    # as recalled, gnark allocates G1.Z with Cardinality entries and uses the first n - 1
    if len(pk.g1_z) >= (1 << pk.log_n) - 1:
        pk.g1_z = pk.g1_z[:(1 << pk.log_n) - 1]
    if (len(pk.a_wire), len(pk.b_wire), len(pk.k_wire)) != (len(pk.g1_a), len(pk.g1_b), len(pk.g1_k)) \
            or len(pk.g2_b) != len(pk.g1_b) or len(pk.g1_z) != (1 << pk.log_n) - 1:
        raise ValueError("point counts do not match the infinity maps / domain")
    return pk


def verifying_key_to_bytes(vk, g1_beta, g1_delta) -> bytes:
    """gnark's VerifyingKey also carries G1.Beta and G1.Delta (this package's does not need them:
    pass the proving key's)."""
    g1 = lambda p: g1_to_bytes(p, compressed=False)
    g2 = lambda p: g2_to_bytes(p, compressed=False)
    return b"".join([g1(vk.g1_alpha), g1(g1_beta), g2(vk.g2_beta), g2(vk.g2_gamma), g1(g1_delta),
                     g2(vk.g2_delta), _pts(vk.g1_k, g1_to_bytes), struct.pack(">II", 0, 0)])


def verifying_key_from_bytes(buf: bytes):
    from .groth16 import VerifyingKey
    vk = VerifyingKey()
    o = 0
    vk.g1_alpha = g1_from_bytes(buf[o:o + 64]); o += 64
    o += 64                                              # G1.Beta
    vk.g2_beta = g2_from_bytes(buf[o:o + 128]); o += 128
    vk.g2_gamma = g2_from_bytes(buf[o:o + 128]); o += 128
    o += 64                                              # G1.Delta
    vk.g2_delta = g2_from_bytes(buf[o:o + 128]); o += 128
    vk.g1_k, o = _read_pts(buf, o, 64, g1_from_bytes, 8)
    a, b = struct.unpack_from(">II", buf, o)
    if a or b:
        raise ValueError("verifying keys with commitments are not supported")
    return vk


# gnark's R1CS.WriteTo is a CBOR encoding of the constraint system's internal tables (blueprints,
# packed instructions, coefficient table, level partition): it is produced and consumed by gnark's
# own solver, which a gnark-side caller keeps (zkmi_prove_witness_batch), so it is
# not restated here.
def save_key(path: str, pk, vk=None):
    """Proving key (and verifying key) in gnark's raw formats, one file each."""
    with open(path, "wb") as f:
        f.write(proving_key_to_bytes(pk))
    if vk is not None:
        with open(path + ".vk", "wb") as f:
            f.write(verifying_key_to_bytes(vk, pk.g1_beta, pk.g1_delta))


def load_key(path: str, n_public: int):
    pk = proving_key_from_bytes(open(path, "rb").read(), n_public)
    try:
        vk = verifying_key_from_bytes(open(path + ".vk", "rb").read())
    except OSError:
        vk = None
    return pk, vk
