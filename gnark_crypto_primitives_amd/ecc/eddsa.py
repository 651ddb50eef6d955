"""EdDSA-Poseidon signature verifier on BabyJubJub, iden3/circomlib compatible.

Mirror of the reference's ecc/bn254/eddsa package: ``Verifier`` / ``NewVerifier`` (verifier.go:
16-37), ``PointToRTE`` (:40-49), ``IsValid`` (:55-88), ``Verify`` (:92-94), ``PublicKey`` /
``Signature`` (types.go:13-22), ``rteB8`` (constants.go:11-18).  The hash is computed on the points
in iden3's TE coordinates, the curve arithmetic runs in gnark's reduced form.
"""
from . import babyjub_native as bjj
from . import format as teformat
from ..std.twistededwards import Curve, Point

# BabyJubJub B8 in reduced twisted-Edwards coordinates = gnark's base point
RTE_B8 = bjj.BASE


class PublicKey:
    def __init__(self, A):
        self.A = A


class Signature:
    def __init__(self, R, S):
        self.R, self.S = R, S


class Verifier:
    def __init__(self, api, hash_fn):
        """hash_fn: a hash.Hash[frontend.Variable] (e.g. hash.Poseidon(api))."""
        self.api = api
        self.curve = Curve(api)
        self.hash_fn = hash_fn

    def PointToRTE(self, p):
        x, y = teformat.FromTEtoRTE(self.api, p.X, p.Y)
        q = Point(x, y)
        self.curve.AssertIsOnCurve(q)
        return q

    def IsValid(self, pub_key, sig, msg):
        api, curve, h = self.api, self.curve, self.hash_fn
        h.Reset()
        h.Write(sig.R.X, sig.R.Y, pub_key.A.X, pub_key.A.Y, msg)
        if not h.WriteSucceeded():
            return 0
        rte_a = self.PointToRTE(pub_key.A)
        rte_r = self.PointToRTE(sig.R)
        left = curve.ScalarMul(Point(*RTE_B8), sig.S)
        r1 = curve.ScalarMul(rte_a, h.Sum())
        r1 = curve.Double(curve.Double(curve.Double(r1)))
        right = curve.Add(r1, rte_r)
        x_valid = api.IsZero(api.Sub(left.X, right.X))
        y_valid = api.IsZero(api.Sub(left.Y, right.Y))
        return api.And(x_valid, y_valid)

    def Verify(self, pub_key, sig, msg):
        self.api.AssertIsEqual(self.IsValid(pub_key, sig, msg), 1)


def NewVerifier(api, hash_fn):
    return Verifier(api, hash_fn)


def sign_native(s, r, msg, hasher):
    """Off-circuit iden3 signature with secret scalar s and nonce r (test data): returns
    (A_te, R8_te, S).  S = r + 8 h s mod order, h = hasher([R8.x, R8.y, A.x, A.y, msg])."""
    a_te = teformat.rte_to_te_native(*bjj.mul(bjj.BASE, s))
    r8_te = teformat.rte_to_te_native(*bjj.mul(bjj.BASE, r))
    hm = hasher([r8_te[0], r8_te[1], a_te[0], a_te[1], msg])
    return a_te, r8_te, (r + hm * 8 * s) % bjj.ORDER
