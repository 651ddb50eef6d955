"""``EncryptedZero`` (reference elgamal/encrypt.go:72-94); ``Encrypt`` lives on Ciphertext."""
from ..std.twistededwards import Curve, Point
from .ciphertext import Ciphertext
from .mul import FixedBaseScalarMulBN254


def EncryptedZero(api, pub_key, k):
    curve = Curve(api)
    curve.AssertIsOnCurve(pub_key)
    c1 = FixedBaseScalarMulBN254(api, k)
    s = curve.ScalarMul(pub_key, k)
    c2 = curve.Add(Point(0, 1), s)
    return Ciphertext(c1, c2)
