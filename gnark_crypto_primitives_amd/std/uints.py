"""Byte and 64-bit word values inside a circuit: what the reference takes from gnark's
``std/math/uints`` (U8, ``uints.New[uints.U64]`` / ``BinaryField.ValueOf``, utils/uints.go:14-28).

gnark range-checks bytes with a log-derivative lookup argument backed by a commitment
(``uints.New`` -> ``rangecheck.New``).  ``BinaryField(api, commit=True)`` does the same: ``ValueOf``
takes the bytes from a hint, asserts the recomposition and hands every byte to the commitment-based
range checker (std/rangecheck.py).  The default (``commit=False``) pins a byte by its eight boolean
wires instead -- cheaper for this repo's boolean Keccak, which needs the bits anyway, and free of the
commitment.  The gadget-level meaning is the same either way: ``ValueOf(v)`` constrains v < 2^64 and
returns its eight bytes little-endian; every U8 is constrained to [0, 256).
"""


_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class U8:
    """uints.U8: ``Val`` is the byte as a field variable; ``bits`` (LSB first) are its boolean
    wires when they are already known, so that bitwise gadgets do not decompose twice."""
    __slots__ = ("Val", "bits")

    def __init__(self, val, bits=None):
        self.Val = val
        self.bits = bits


def NewU8(v):
    """uints.NewU8: a constant byte."""
    if not 0 <= int(v) < 256:
        raise ValueError("byte out of range")
    return U8(int(v))


class BinaryField:
    """uints.BinaryField[U64] (``uints.New[uints.U64](api)``)."""
    def __init__(self, api, word_bytes=8, commit=False):
        self.api = api
        self.word_bytes = word_bytes
        self.rchecker = None
        if commit:
            from . import rangecheck
            self.rchecker = rangecheck.New(api)

    def Bits(self, b):
        """The eight boolean wires of a byte, LSB first (decomposed once, then cached)."""
        if b.bits is None:
            b.bits = self.api.ToBinary(b.Val, 8)
        return b.bits

    def ByteFromBits(self, bits):
        assert len(bits) == 8
        return U8(self.api.FromBinary(*bits), list(bits))

    def ByteValueOf(self, v):
        """uints.BinaryField.ByteValueOf: v as one range-checked byte."""
        if self.rchecker is not None:
            self.rchecker.Check(v, 8)
            return U8(v)
        bits = self.api.ToBinary(v, 8)
        return U8(v, bits)

    def ValueOf(self, v):
        """uints.BinaryField.ValueOf: v < 2^(8*word_bytes) as bytes, least significant first."""
        if self.rchecker is not None:
            # gnark: bytes from the toBytes hint, each through ByteValueOf, recomposition asserted
            bts = self.api.NewHintLimbs(v, 8, self.word_bytes)
            word = [self.ByteValueOf(b) for b in bts]
            self.api.AssertIsEqual(v, self.ToValue(word))
            return word
        bits = self.api.ToBinary(v, 8 * self.word_bytes)
        return [self.ByteFromBits(bits[8 * i:8 * i + 8]) for i in range(self.word_bytes)]

    def ToValue(self, word):
        """uints.BinaryField.ToValue: little-endian bytes back to one variable."""
        api = self.api
        acc = 0
        for i, b in enumerate(word):
            acc = api.Add(acc, api.Mul(b.Val, 1 << (8 * i)))
        return acc

    def AssertEq(self, a, b):
        self.api.AssertIsEqual(a.Val, b.Val)

    # ---- byte-wise logic the way gnark's uints does it (commit mode): lookup tables + range checks --
    def _table(self, op):
        from . import logderivprecomp
        if self.rchecker is None:
            raise ValueError("byte-wise Xor / And / Not / Lrot need BinaryField(api, commit=True)")
        return logderivprecomp.New(self.api, op)

    def _two(self, op, a, b):
        return [U8(self._table(op).Query(x.Val, y.Val)) for x, y in zip(a, b)]

    def Xor(self, a, *rest):
        """uints.BinaryField.Xor: bytes through the 2^16-row XOR table, operand by operand."""
        from ..frontend.api import OP_BXOR
        for b in rest:
            a = self._two(OP_BXOR, a, b)
        return list(a)

    def And(self, a, *rest):
        from ..frontend.api import OP_BAND
        for b in rest:
            a = self._two(OP_BAND, a, b)
        return list(a)

    def Not(self, a):
        """uints.BinaryField.Not: XOR with 0xff through the table (as gnark does)."""
        from ..frontend.api import OP_BXOR
        return self._two(OP_BXOR, a, [U8(0xff)] * len(a))

    def Lrot(self, a, c):
        """uints.BinaryField.Lrot: rotate the word left by c bits.  Whole bytes move by renaming; the
        remaining c % 8 bits split every byte with bitslice.Partition (a hint for the low part, the
        high part follows linearly; both range-checked) and stitch neighbours together."""
        n = len(a)
        c %= 8 * n
        shift_bl, shift_bt = c // 8, c % 8
        if shift_bt == 0:
            return [a[(i - shift_bl) % n] for i in range(n)]
        rev = 8 - shift_bt
        parts = []
        for b in a:
            lo = self.api.NewHintLimbs(b.Val, rev, 1)[0]                  # low `rev` bits of the byte
            hi = self.api.Mul(self.api.Sub(b.Val, lo), pow(1 << rev, -1, _R))   # (b - lo) / 2^rev
            self.rchecker.Check(lo, rev)
            self.rchecker.Check(hi, shift_bt)
            parts.append((lo, hi))
        out = [None] * n
        for i in range(n):
            # new byte = low part shifted up + high part of the byte below
            out[(i + shift_bl) % n] = U8(self.api.Add(self.api.Mul(parts[i][0], 1 << shift_bt),
                                                      parts[(i - 1) % n][1]))
        return out
