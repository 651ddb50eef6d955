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
