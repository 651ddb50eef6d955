"""The slice of gnark's ``std/math/emulated`` the reference's byte helpers touch: an element of a
foreign field as little-endian limbs (``Element.Limbs``), ``ValueOf`` for assignments and the two
parameter sets the reference instantiates (Secp256k1Fp/Fr: 4 limbs x 64 bits).  Emulated
arithmetic itself (Reduce, Mul, ...) is not on this path and is not restated."""


class FieldParams:
    def __init__(self, name, modulus, nb_limbs=4, bits_per_limb=64):
        self.name, self.modulus = name, modulus
        self.nb_limbs, self.bits_per_limb = nb_limbs, bits_per_limb

    def NbLimbs(self):
        return self.nb_limbs

    def BitsPerLimb(self):
        return self.bits_per_limb

    def Modulus(self):
        return self.modulus


Secp256k1Fp = FieldParams("Secp256k1Fp", 2**256 - 2**32 - 977)
Secp256k1Fr = FieldParams(
    "Secp256k1Fr", 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141)
BLS12377Fr = FieldParams(
    "BLS12377Fr", 0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001)


class Element:
    """emulated.Element[T]: ``Limbs`` least significant first."""
    def __init__(self, limbs, params=Secp256k1Fp):
        self.Limbs = list(limbs)
        self.params = params


def limbs_of(x: int, params=Secp256k1Fp):
    """emulated.ValueOf: the integer as nb_limbs limbs of bits_per_limb bits (assignment side)."""
    x %= params.modulus
    mask = (1 << params.bits_per_limb) - 1
    return [(x >> (params.bits_per_limb * i)) & mask for i in range(params.nb_limbs)]
