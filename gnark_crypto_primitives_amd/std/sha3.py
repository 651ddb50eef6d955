"""Keccak-f[1600] sponge in R1CS: ``sha3.NewLegacyKeccak256`` of gnark's std/hash/sha3, the hash
behind ecdsa.DeriveAddress (reference ecc/secp256k1/ecdsa/address.go:27-32).

The state is 25 lanes x 64 boolean wires.  Per round: theta 3 200 XOR constraints (column parities
as XOR chains, D = C[x-1] ^ rot(C[x+1], 1), A ^= D), rho and pi are wire renamings, chi is one AND
and one XOR per bit (3 200), iota XORs a constant (linear, free): 6 400 constraints per round,
153 600 per permutation.
"""
from .uints import U8, BinaryField

RATE_256 = 136

# round constants and rotation offsets generated from the Keccak specification's LFSR / (x, y) walk
def _round_constants():
    rc, r = [], 1
    for _ in range(24):
        c = 0
        for j in range(7):
            if r & 1:
                c |= 1 << ((1 << j) - 1)
            r <<= 1
            if r & 0x100:
                r ^= 0x171
        rc.append(c)
    return rc


def _rotations():
    rot = [[0] * 5 for _ in range(5)]
    x, y = 1, 0
    for t in range(24):
        rot[x][y] = ((t + 1) * (t + 2) // 2) % 64
        x, y = y, (2 * x + 3 * y) % 5
    return rot


RC = _round_constants()
ROT = _rotations()


def _rol(lane, n):
    """bit i of the result = bit (i - n) of the lane (rotation towards the high end)."""
    n %= 64
    return lane[-n:] + lane[:-n] if n else list(lane)


def permute(api, A):
    """Keccak-f[1600] on A[x][y] = list of 64 boolean variables (LSB first)."""
    for rnd in range(24):
        # theta
        C = []
        for x in range(5):
            col = A[x][0]
            for y in range(1, 5):
                col = [api.Xor(p, q) for p, q in zip(col, A[x][y])]
            C.append(col)
        D = [[api.Xor(p, q) for p, q in zip(C[(x - 1) % 5], _rol(C[(x + 1) % 5], 1))]
             for x in range(5)]
        A = [[[api.Xor(p, q) for p, q in zip(A[x][y], D[x])] for y in range(5)] for x in range(5)]
        # rho + pi
        B = [[None] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], ROT[x][y])
        # chi
        A = [[[api.Xor(B[x][y][i], api.And(api.Xor(B[(x + 1) % 5][y][i], 1), B[(x + 2) % 5][y][i]))
               for i in range(64)] for y in range(5)] for x in range(5)]
        # iota: XOR with a constant bit is 1 - b
        rc = RC[rnd]
        A[0][0] = [api.Xor(b, 1) if (rc >> i) & 1 else b for i, b in enumerate(A[0][0])]
    return A


class LegacyKeccak256:
    """hash.BinaryFixedLengthHasher as returned by sha3.NewLegacyKeccak256(api): ``Write`` takes
    uints.U8 values, ``Sum`` returns the 32 digest bytes.  Padding 0x01 .. 0x80 (pre-NIST)."""
    def __init__(self, api):
        self.api = api
        self.bf = BinaryField(api)
        self.data = []

    def Write(self, data):
        self.data.extend(data)

    def Size(self):
        return 32

    def Sum(self):
        api, bf = self.api, self.bf
        msg = list(self.data)
        pad = RATE_256 - len(msg) % RATE_256
        tail = [0] * pad
        tail[0] |= 0x01
        tail[-1] |= 0x80
        msg += [U8(t, [(t >> i) & 1 for i in range(8)]) for t in tail]
        A = [[[0] * 64 for _ in range(5)] for _ in range(5)]
        for off in range(0, len(msg), RATE_256):
            block = msg[off:off + RATE_256]
            for k in range(RATE_256 // 8):
                lane = [b for byte in block[8 * k:8 * k + 8] for b in bf.Bits(byte)]
                x, y = k % 5, k // 5
                A[x][y] = [api.Xor(p, q) for p, q in zip(A[x][y], lane)]
            A = permute(api, A)
        out = []
        for k in range(4):
            lane = A[k % 5][k // 5]
            out += [bf.ByteFromBits(lane[8 * i:8 * i + 8]) for i in range(8)]
        return out


def NewLegacyKeccak256(api):
    return LegacyKeccak256(api)


# ---- the byte-wise form: gnark's std/permutation/keccakf over uints.U64 [UPSTREAM-RECALL] ---------------
def permute_bytes(bf, A):
    """Keccak-f[1600] on A[x][y] = uints.U64 (eight range-checked bytes, least significant first),
    with gnark's uints operations: Xor / And / Not are lookups into 2^16-row tables, rotations by
    whole bytes are renamings, the other rotations split bytes with range checks (uints.Lrot).
    Per round: 76 Xor64 + 25 And64 + 25 Not64 = 1 008 byte lookups, 27 bit-level rotations."""
    for rnd in range(24):
        C = [bf.Xor(A[x][0], A[x][1], A[x][2], A[x][3], A[x][4]) for x in range(5)]
        D = [bf.Xor(C[(x - 1) % 5], bf.Lrot(C[(x + 1) % 5], 1)) for x in range(5)]
        A = [[bf.Xor(A[x][y], D[x]) for y in range(5)] for x in range(5)]
        B = [[None] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = bf.Lrot(A[x][y], ROT[x][y])
        A = [[bf.Xor(B[x][y], bf.And(bf.Not(B[(x + 1) % 5][y]), B[(x + 2) % 5][y]))
              for y in range(5)] for x in range(5)]
        rc = [U8((RC[rnd] >> (8 * i)) & 0xff) for i in range(8)]
        A[0][0] = bf.Xor(A[0][0], rc)
    return A


class LegacyKeccak256Bytes:
    """sha3.NewLegacyKeccak256 as gnark builds it for an R1CS: the sponge over uints.U64 lanes."""
    def __init__(self, api):
        self.api = api
        self.bf = BinaryField(api, commit=True)
        self.data = []

    def Write(self, data):
        self.data.extend(data)

    def Size(self):
        return 32

    def Sum(self):
        bf = self.bf
        msg = list(self.data)
        pad = RATE_256 - len(msg) % RATE_256
        tail = [0] * pad
        tail[0] |= 0x01
        tail[-1] |= 0x80
        msg += [U8(t) for t in tail]
        A = [[[U8(0)] * 8 for _ in range(5)] for _ in range(5)]
        for off in range(0, len(msg), RATE_256):
            block = msg[off:off + RATE_256]
            for k in range(RATE_256 // 8):
                x, y = k % 5, k // 5
                A[x][y] = bf.Xor(A[x][y], block[8 * k:8 * k + 8])
            A = permute_bytes(bf, A)
        out = []
        for k in range(4):
            out += A[k % 5][k // 5]
        return out


def NewLegacyKeccak256Bytes(api):
    return LegacyKeccak256Bytes(api)
