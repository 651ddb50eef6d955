"""Off-circuit counterparts used to build assignments (what the reference takes from go-ethereum
in testutil/utils.go:44-93): Keccak-256 with the legacy padding, secp256k1 public keys, addresses."""
from ...std.sha3 import RC, ROT, RATE_256

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
     0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)
M64 = (1 << 64) - 1


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & M64 if n else v


def keccak_f(A):
    for rc in RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], ROT[x][y])
        A = [[B[x][y] ^ (~B[(x + 1) % 5][y] & M64 & B[(x + 2) % 5][y]) for y in range(5)]
             for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data: bytes) -> bytes:
    msg = bytearray(data)
    pad = RATE_256 - len(msg) % RATE_256
    tail = bytearray(pad)
    tail[0] |= 0x01
    tail[-1] |= 0x80
    msg += tail
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), RATE_256):
        for k in range(RATE_256 // 8):
            A[k % 5][k // 5] ^= int.from_bytes(msg[off + 8 * k:off + 8 * k + 8], "little")
        A = keccak_f(A)
    return b"".join(A[k % 5][k // 5].to_bytes(8, "little") for k in range(4))


def _add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % P == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, P) % P
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, P) % P
    x = (lam * lam - p[0] - q[0]) % P
    return x, (lam * (p[0] - x) - p[1]) % P


def public_key(priv: int):
    acc, base, k = None, G, priv % N
    while k:
        if k & 1:
            acc = _add(acc, base)
        base = _add(base, base)
        k >>= 1
    return acc


def address(pub) -> int:
    """Ethereum address (as an integer) of an affine public key: low 20 bytes of
    Keccak-256(X || Y), both 32 bytes big-endian."""
    h = keccak256(pub[0].to_bytes(32, "big") + pub[1].to_bytes(32, "big"))
    return int.from_bytes(h[12:], "big")
