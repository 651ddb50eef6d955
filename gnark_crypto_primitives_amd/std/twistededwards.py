"""Twisted-Edwards curve gadget over the circuit's native field, BN254 instance (BabyJubJub).

Restates gnark's ``std/algebra/native/twistededwards`` as the reference uses it
(``twistededwards.NewEdCurve(api, ecc_tweds.BN254)``: elgamal/ciphertext.go:25-31,
elgamal/encrypt.go:43, elgamal/mul.go:80) [UPSTREAM-RECALL]: ``Add`` (unified, 6 constraints),
``Double``, ``Neg``, ``AssertIsOnCurve``, ``ScalarMul`` (2-bit windows with Lookup2 over
{0, P, 2P, 3P}, MSB first).
"""
from ..ecc import babyjub_native as bjj


class Point:
    __slots__ = ("X", "Y")

    def __init__(self, X, Y):
        self.X, self.Y = X, Y


class Curve:
    """``twistededwards.NewEdCurve(api, BN254)``."""

    def __init__(self, api):
        self.api = api
        self.A, self.D = bjj.A, bjj.D
        self.base = bjj.BASE
        self.order = bjj.ORDER

    def Add(self, p1, p2):
        api = self.api
        u1 = api.Sub(p1.Y, api.Mul(p1.X, self.A))
        u2 = api.Add(p2.X, p2.Y)
        u = api.Mul(u1, u2)
        v0 = api.Mul(p2.Y, p1.X)
        v1 = api.Mul(p2.X, p1.Y)
        v2 = api.Mul(self.D, v0, v1)
        px = api.DivUnchecked(api.Add(v0, v1), api.Add(1, v2))
        py = api.Add(api.Sub(api.Mul(self.A, v0), v1), u)
        py = api.DivUnchecked(py, api.Sub(1, v2))
        return Point(px, py)

    def Double(self, p1):
        api = self.api
        u = api.Mul(p1.X, p1.Y)
        v = api.Mul(p1.X, p1.X)
        w = api.Mul(p1.Y, p1.Y)
        n1 = api.Mul(2, u)
        av = api.Mul(v, self.A)
        n2 = api.Sub(w, av)
        d1 = api.Add(w, av)
        d2 = api.Sub(2, d1)
        return Point(api.DivUnchecked(n1, d1), api.DivUnchecked(n2, d2))

    def Neg(self, p1):
        return Point(self.api.Neg(p1.X), p1.Y)

    def AssertIsOnCurve(self, p1):
        api = self.api
        xx = api.Mul(p1.X, p1.X)
        yy = api.Mul(p1.Y, p1.Y)
        lhs = api.Add(api.Mul(xx, self.A), yy)
        rhs = api.Add(api.Mul(api.Mul(xx, self.D), yy), 1)
        api.AssertIsEqual(lhs, rhs)

    def ScalarMul(self, p1, scalar):
        api = self.api
        b = api.ToBinary(scalar, 254)
        A = self.Double(p1)
        B = self.Add(A, p1)
        n = len(b) - 1
        res = Point(api.Lookup2(b[n], b[n - 1], 0, A.X, p1.X, B.X),
                    api.Lookup2(b[n], b[n - 1], 1, A.Y, p1.Y, B.Y))
        for i in range(n - 2, 0, -2):
            res = self.Double(self.Double(res))
            tmp = Point(api.Lookup2(b[i], b[i - 1], 0, A.X, p1.X, B.X),
                        api.Lookup2(b[i], b[i - 1], 1, A.Y, p1.Y, B.Y))
            res = self.Add(res, tmp)
        if n % 2 == 0:
            res = self.Double(res)
            tmp = self.Add(res, p1)
            res = Point(api.Select(b[0], tmp.X, res.X), api.Select(b[0], tmp.Y, res.Y))
        return res
