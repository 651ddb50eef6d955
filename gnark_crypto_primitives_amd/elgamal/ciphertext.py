"""ElGamal ciphertext over BabyJubJub and the Chaum-Pedersen decryption proof.

Mirror of the reference's elgamal/ciphertext.go: ``Ciphertext`` (:12), ``NewCiphertext`` (:16),
``Add`` (:24), ``Neg`` (:37), ``AssertDecrypt`` (:50-67), ``AssertIsEqual`` (:70), ``IsEqual``
(:79), ``Select`` (:89), ``Serialize`` (:98), ``DecryptionProof`` (:109), ``Verify`` (:124-168),
``hashPointsToScalar`` (:173-184).  ``Encrypt`` (encrypt.go:42-64) is a method here too.
"""
from ..std.twistededwards import Curve, Point
from .mul import FixedBaseScalarMulBN254


class Ciphertext:
    def __init__(self, C1=None, C2=None):
        self.C1 = C1 if C1 is not None else Point(0, 1)
        self.C2 = C2 if C2 is not None else Point(0, 1)

    def Add(self, api, x, y):
        curve = Curve(api)
        self.C1 = curve.Add(x.C1, y.C1)
        self.C2 = curve.Add(x.C2, y.C2)
        return self

    def Neg(self, api, x):
        curve = Curve(api)
        self.C1 = curve.Neg(x.C1)
        self.C2 = curve.Neg(x.C2)
        return self

    def AssertDecrypt(self, api, priv_key, m):
        curve = Curve(api)
        curve.AssertIsOnCurve(self.C1)
        curve.AssertIsOnCurve(self.C2)
        S = curve.ScalarMul(self.C1, priv_key)
        M = FixedBaseScalarMulBN254(api, m)
        m_prime = curve.Add(self.C2, curve.Neg(S))
        api.AssertIsEqual(m_prime.X, M.X)
        api.AssertIsEqual(m_prime.Y, M.Y)

    def AssertIsEqual(self, api, x):
        api.AssertIsEqual(self.C1.X, x.C1.X)
        api.AssertIsEqual(self.C1.Y, x.C1.Y)
        api.AssertIsEqual(self.C2.X, x.C2.X)
        api.AssertIsEqual(self.C2.Y, x.C2.Y)

    def IsEqual(self, api, x):
        return api.Mul(api.IsZero(api.Sub(self.C1.X, x.C1.X)),
                       api.IsZero(api.Sub(self.C1.Y, x.C1.Y)),
                       api.IsZero(api.Sub(self.C2.X, x.C2.X)),
                       api.IsZero(api.Sub(self.C2.Y, x.C2.Y)))

    def Select(self, api, b, i1, i2):
        self.C1 = Point(api.Select(b, i1.C1.X, i2.C1.X), api.Select(b, i1.C1.Y, i2.C1.Y))
        self.C2 = Point(api.Select(b, i1.C2.X, i2.C2.X), api.Select(b, i1.C2.Y, i2.C2.Y))
        return self

    def Serialize(self):
        return [self.C1.X, self.C1.Y, self.C2.X, self.C2.Y]

    def Encrypt(self, api, pub_key, k, m):
        """encrypt.go:42-64: C1 = [k]G, C2 = [m]G + [k]P."""
        curve = Curve(api)
        curve.AssertIsOnCurve(pub_key)
        self.C1 = FixedBaseScalarMulBN254(api, k)
        s = curve.ScalarMul(pub_key, k)
        m_point = FixedBaseScalarMulBN254(api, m)
        self.C2 = curve.Add(m_point, s)
        return self


def NewCiphertext():
    return Ciphertext()


def hashPointsToScalar(api, hFn, *points):
    coords = []
    for p in points:
        coords += [p.X, p.Y]
    return hFn(api, *coords)


class DecryptionProof:
    """Non-interactive Chaum-Pedersen proof that C2 - [m]G and C1 share the discrete log of P."""

    def __init__(self, A1, A2, Z):
        self.A1, self.A2, self.Z = A1, A2, Z

    def Verify(self, api, hFn, pubkey, ciphertext, msg):
        curve = Curve(api)
        for pt in (pubkey, ciphertext.C1, ciphertext.C2, self.A1, self.A2):
            curve.AssertIsOnCurve(pt)
        M = FixedBaseScalarMulBN254(api, msg)
        D = curve.Add(ciphertext.C2, curve.Neg(M))
        # the public key is hashed twice: elgamal/ciphertext.go:146
        E = hashPointsToScalar(api, hFn, pubkey, pubkey, ciphertext.C1, D, self.A1, self.A2)
        zG = FixedBaseScalarMulBN254(api, self.Z)
        a1_plus_ep = curve.Add(self.A1, curve.ScalarMul(pubkey, E))
        api.AssertIsEqual(a1_plus_ep.X, zG.X)
        api.AssertIsEqual(a1_plus_ep.Y, zG.Y)
        zC1 = curve.ScalarMul(ciphertext.C1, self.Z)
        a2_plus_ed = curve.Add(self.A2, curve.ScalarMul(D, E))
        api.AssertIsEqual(a2_plus_ed.X, zC1.X)
        api.AssertIsEqual(a2_plus_ed.Y, zC1.Y)
