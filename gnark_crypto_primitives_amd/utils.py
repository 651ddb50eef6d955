"""The reference's ``utils`` package: hasher adaptors (utils/hashers.go:10-27), byte helpers
(utils/uints.go, utils/bytes.go) and scalar packing (utils/utils.go)."""
from .hash import poseidon as _poseidon
from .std import emulated as _emulated
from .std.uints import U8, BinaryField, NewU8


def PoseidonHasher(api, *data):
    """utils.PoseidonHasher (utils/hashers.go:25-27)."""
    return _poseidon.Hash(api, *data)


def PoseidonMultiHasher(api, *data):
    """``poseidon.MultiHash`` as a utils.Hasher (the reference's elgamal tests use
    ``HashFn = poseidon.MultiHash``, elgamal/ciphertext_test.go:272)."""
    return _poseidon.MultiHash(api, *data)


# ---- utils/uints.go -------------------------------------------------------------------------------
def ElemToU8(api, elem, commit=False):
    """utils.ElemToU8 (utils/uints.go:14-28): every limb through ``bf.ValueOf`` (eight bytes,
    least significant first), limbs in order.  commit: range-check the bytes the way gnark's
    ``uints.New`` does, through the commitment-based checker (std/rangecheck.py)."""
    bf = BinaryField(api, commit=commit)
    res = []
    for limb in elem.Limbs:
        res.extend(bf.ValueOf(limb))
    return res


def U8ToVar(api, u8):
    """utils.U8ToVar (utils/uints.go:33-46): big-endian bytes to one variable."""
    n = len(u8)
    acc = 0
    for i, b in enumerate(u8):
        acc = api.Add(acc, api.Mul(b.Val, 256 ** (n - 1 - i)))
    return acc


def U8ToElem(api, u8s, params=_emulated.Secp256k1Fp):
    """utils.U8ToElem (utils/uints.go:50-91): inverse of ElemToU8; pads with zero bytes or
    truncates to nbLimbs * 8 bytes."""
    total = params.nb_limbs * 8
    u8s = list(u8s)[:total] + [NewU8(0)] * max(0, total - len(u8s))
    limbs = []
    for i in range(params.nb_limbs):
        acc = 0
        for j in range(8):
            acc = api.Add(acc, api.Mul(u8s[8 * i + j].Val, 256 ** j))
        limbs.append(acc)
    return _emulated.Element(limbs, params)


def varToLimbsOfBits(api, v, n_limbs, nb_bits):
    """utils.varToLimbsOfBits (utils/uints.go:128-139)."""
    bits = api.ToBinary(v, nb_bits * n_limbs)
    return [api.FromBinary(*bits[i * nb_bits:(i + 1) * nb_bits]) for i in range(n_limbs)]


def SwapEndianness(u8):
    """utils.SwapEndianness (utils/uints.go:117-123)."""
    return list(reversed(u8))


def VarToU8(api, v):
    """utils.VarToU8 (utils/uints.go:97-113): 4 x 64-bit limbs -> bytes -> reversed (big-endian,
    32 bytes)."""
    bf = BinaryField(api)
    u8 = []
    for limb in varToLimbsOfBits(api, v, 4, 64):
        u8.extend(bf.ValueOf(limb))
    return SwapEndianness(u8)


# ---- utils/bytes.go -------------------------------------------------------------------------------
class Bytes(list):
    """utils.Bytes (utils/bytes.go:13-60): a slice of uints.U8."""
    def AssertIsEqual(self, api, other):
        if len(self) != len(other):
            api.AssertIsEqual(0, 1)
            return
        for a, b in zip(self, other):
            api.AssertIsEqual(a.Val, b.Val)

    def IsEqual(self, api, other):
        if len(self) != len(other):
            return 0
        matches = 0
        for a, b in zip(self, other):
            matches = api.Add(matches, api.IsZero(api.Sub(a.Val, b.Val)))
        return api.IsZero(api.Sub(matches, len(self)))

    def ToVar(self, api):
        return U8ToVar(api, self)

    def Values(self):
        return [b.Val for b in self]


def BytesFromElement(api, e):
    return Bytes(ElemToU8(api, e))


def BytesFromVariable(api, v):
    return Bytes(VarToU8(api, v))


def BytesFromBigInt(b: int, fixed_len: int):
    """utils.BytesFromBigInt (utils/bytes.go:66-76): big-endian bytes of b, truncated to
    fixed_len, then zero-filled at the END."""
    raw = b.to_bytes((b.bit_length() + 7) // 8, "big")[:fixed_len]
    return Bytes([NewU8(x) for x in raw] + [NewU8(0)] * (fixed_len - len(raw)))


def BytesFromString(s: str, fixed_len: int):
    """utils.BytesFromString (utils/bytes.go:81-83); leading NUL bytes vanish as in big.Int."""
    return BytesFromBigInt(int.from_bytes(s.encode(), "big"), fixed_len)


# ---- utils/utils.go -------------------------------------------------------------------------------
def PackScalarToVar(api, s):
    """utils.PackScalarToVar (utils/utils.go:14-32): ``field.Reduce`` (limb widths enforced; an
    element with lazy additions behind it is reduced by a product check), then sum limb_i 2^(64 i)."""
    s = _emulated.NewField(api, s.params).Reduce(s)
    acc = 0
    for i, limb in enumerate(s.Limbs):
        acc = api.Add(acc, api.Mul(limb, 1 << (s.params.bits_per_limb * i)))
    return acc


def UnpackVarToScalar(api, v, params=_emulated.Secp256k1Fr):
    """utils.UnpackVarToScalar (utils/utils.go:40-51)."""
    return _emulated.Element(varToLimbsOfBits(api, v, params.nb_limbs, params.bits_per_limb),
                             params)


def StrictCmp(api, a, b):
    """utils.StrictCmp (utils/utils.go:57-59): 1 if a != b else 0."""
    return api.Select(api.IsZero(api.Sub(a, b)), 0, 1)
