"""gnark ``std/math/emulated`` (reference users: hash/emulated/bn254/poseidon,
tree/smt/emulated, utils.U8ToElem) on the CPU side [UPSTREAM-RECALL; parity UNPINNED: gnark is not
installed and the reference holds no vector of an emulated circuit's constraint system].  What is
checked: the frontend's two witness-program interpreters and the C oracle's gnark-style solver (its
own big-integer division, oracle/c/zkref_prove.inc kind 7) agree on every wire and row; satisfied
exactly when the emulated statement is true; and the reference's own test of the emulated Poseidon
-- hash/emulated/bn254/poseidon/poseidon_test.go:45-78: Hash(1, 2, 3) equals the native permutation's
value -- holds on this frontend."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16, verify
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import (Public, Secret, emul_unit,
                                                          from_mont_array, to_mont_array)
from gnark_crypto_primitives_amd.hash import emulated_poseidon, poseidon_native
from gnark_crypto_primitives_amd.std import emulated as em
from oracle import cref
from tests import helpers as H


class ArithCircuit:
    """z = ((x + y)(x - y))^2 + 7 over the emulated field, asserted twice (AssertIsEqual and
    IsZero of the difference); flag = [x == y]; s = flag ? x : z, compared bit by bit with S"""
    Z = Public(4)
    Flag = Public()
    X = Secret(4)
    Y = Secret(4)
    S = Secret(4)

    def __init__(self, params):
        self.params = params

    def define(self, api):
        f = em.NewField(api, self.params)
        x, y, z, s = (em.Element(v, self.params) for v in (self.X, self.Y, self.Z, self.S))
        t = f.Mul(f.Add(x, y), f.Sub(x, y))
        t = f.Add(f.Mul(t, t), 7)
        f.AssertIsEqual(t, z)
        api.AssertIsEqual(f.IsZero(f.Sub(t, z)), 1)
        flag = f.IsZero(f.Sub(x, y))
        api.AssertIsEqual(flag, self.Flag)
        sel = f.Select(flag, x, t)
        for p, q in zip(f.ToBits(sel), f.ToBits(s)):
            api.AssertIsEqual(p, q)

    def assignment(self, x, y, z=None, flag=None, s=None):
        p = self.params.modulus
        zz = (pow((x * x - y * y) % p, 2, p) + 7) % p
        fl = int(x % p == y % p)
        return {"X": em.ValueOf(x, self.params), "Y": em.ValueOf(y, self.params),
                "Z": em.ValueOf(zz if z is None else z, self.params),
                "Flag": fl if flag is None else flag,
                "S": em.ValueOf((x if fl else zz) if s is None else s, self.params)}


def _mul(g, s):
    return cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)


def test_emul_unit_is_divmod():
    rng = random.Random(3)
    p = em.Secp256k1Fp.modulus
    consts = [0, 5] + [(p >> (64 * i)) & (2**64 - 1) for i in range(4)]
    for _ in range(50):
        a = [rng.randrange(1 << rng.choice((1, 64, 100, 183))) for _ in range(rng.randrange(1, 5))]
        b = [rng.randrange(1 << rng.choice((1, 64, 70))) for _ in range(rng.randrange(1, 5))]
        ai = sum(v << (64 * i) for i, v in enumerate(a))
        bi = sum(v << (64 * i) for i, v in enumerate(b))
        out = emul_unit(a + b, 12 | len(a) << 8 | 2 << 12, consts)
        k = sum(v << (64 * i) for i, v in enumerate(out[:8]))
        r = sum(v << (64 * i) for i, v in enumerate(out[8:]))
        assert (k, r) == divmod(ai * bi, p) or ai * bi // p >= 1 << 512


@pytest.mark.parametrize("params", [em.BN254Fr, em.Secp256k1Fp], ids=lambda p: p.name)
def test_arithmetic_circuit_three_solvers(params):
    circ = ArithCircuit(params)
    cc = compile_circuit(circ)
    assert len(cc.commitments) == 1
    p = params.modulus
    rng = random.Random(7)
    x, y = rng.randrange(p), rng.randrange(p)
    rh = cref.R1csHandle(cc)
    pk, vk, _ = groth16.setup(cc, 21, _mul)
    ch = cref.CommitKeysHandle(pk)
    cases = [(circ.assignment(x, y), True),
             (circ.assignment(x, x), True),                          # flag = 1, select takes x
             (circ.assignment(p - 1, 1), True),
             (circ.assignment(0, 0), True),
             (circ.assignment(x, y, z=5), False),                    # wrong product
             (circ.assignment(x, y, flag=1), False),
             (circ.assignment(x, y, s=x), False)]
    # a remainder that is not reduced: z + p as limbs (fits 256 bits for BN254's r only)
    zz = (pow((x * x - y * y) % p, 2, p) + 7) % p
    if zz + p < 1 << 254 and params is em.BN254Fr:
        a = circ.assignment(x, y)
        a["Z"] = [((zz + p) >> (64 * i)) & (2**64 - 1) for i in range(4)]
        cases.append((a, True))          # AssertIsEqual is modular: an unreduced public value passes
    real = groth16.commit_fn(pk)
    for n, (asg, ok) in enumerate(cases):
        vec = cc.assignment_vector(asg)
        rc, ow, oa, ob, oc, _ = cref.r1cs_solve_ex(rh, ch, to_mont_array(vec))
        assert (rc == 0) == ok
        ow = from_mont_array(ow)
        # the interpreters get the oracle's challenge (None: the SHA-256 stand-in) -- except once,
        # where the Python Pedersen commitment + hash_to_field must produce that very value
        cc.commit_fn = None if not ok else real if n == 0 else \
            (lambda idx, hashed, committed: ow[cc.commitments[idx]["wire"]])
        w, a, b, c = cc.run_program(vec)
        st = cc.last_status
        w2, a2, b2, c2 = cc.run_vprogram(vec)
        cc.commit_fn = None
        assert (st == 0) == ok and (cc.last_status == 0) == ok
        assert w2 == w and (a2, b2, c2) == (a, b, c)
        assert cc.is_satisfied(w)[0] == ok
        if ok:
            assert ow == w
            assert from_mont_array(oa) == a and from_mont_array(ob) == b and from_mont_array(oc) == c
    # and a proof of it verifies
    vec = cc.assignment_vector(cases[0][0])
    inp = to_mont_array(vec).reshape(1, -1, 4)
    proofs, coms, poks, status, _ = cref.groth16_prove_batch_ex(
        rh, cref.PkHandle(pk), ch, inp, to_mont_array([3, 4]).reshape(1, 2, 4))
    pub = vec[:cc.n_public - 1]
    assert not status.any() and verify.verify(vk, pub, proofs[0], coms[0], poks[0])
    assert not verify.verify(vk, [pub[0] ^ 1] + pub[1:], proofs[0], coms[0], poks[0])


def test_mux_lookup2_and_pack():
    """Field.Mux / Lookup2 (tree/smt/emulated/utils.go:23-34 mux2 / mux3) and utils.PackScalarToVar
    (utils/utils.go:14-32) on an element with lazy additions behind it"""
    from gnark_crypto_primitives_amd import utils
    params = em.BN254Fr
    p = params.modulus

    class C:
        Packed = Public()
        Sel = Secret()
        A = Secret(4)
        B = Secret(4)
        Out = Secret(4)

        def define(self, api):
            f = em.NewField(api, params)
            a, b = em.Element(self.A, params), em.Element(self.B, params)
            z = f.Zero()
            m = f.Mux(self.Sel, z, a, b, f.Add(a, b), a, f.Mul(a, b))     # six inputs
            f.AssertIsEqual(m, em.Element(self.Out, params))
            bits = api.ToBinary(self.Sel, 3)
            l2 = f.Lookup2(bits[0], bits[1], z, a, b, f.Add(a, b))
            api.AssertIsEqual(utils.PackScalarToVar(api, f.Add(l2, l2)), self.Packed)

    cc = compile_circuit(C())
    a, b = 0x1234567 << 200, p - 5
    want = [0, a, b, (a + b) % p, a, a * b % p]
    for sel in range(6):
        asg = {"Packed": 2 * want[sel & 3] % p, "Sel": sel, "A": em.ValueOf(a, params),
               "B": em.ValueOf(b, params), "Out": em.ValueOf(want[sel], params)}
        w, *_ = cc.run_vprogram(cc.assignment_vector(asg))
        assert cc.last_status == 0 and cc.is_satisfied(w)[0], sel
        asg["Out"] = em.ValueOf(want[sel] + 1, params)
        cc.run_vprogram(cc.assignment_vector(asg))
        assert cc.last_status != 0
    asg = {"Packed": 0, "Sel": 6, "A": em.ValueOf(a, params), "B": em.ValueOf(b, params),
           "Out": em.ValueOf(0, params)}
    cc.run_vprogram(cc.assignment_vector(asg))
    assert cc.last_status != 0          # selector beyond the inputs


def test_lazy_additions_force_reductions():
    """a long chain of additions and subtractions grows the overflow until Mul / Add must reduce"""
    params = em.BN254Fr

    class Chain:
        Out = Public(4)
        X = Secret(4)

        def define(self, api):
            f = em.NewField(api, params)
            x = em.Element(self.X, params)
            acc = x
            for i in range(130):
                acc = f.Sub(f.Add(acc, acc), x) if i % 3 else f.Add(acc, f.Mul(acc, 3))
            f.AssertIsEqual(f.Mul(acc, acc), em.Element(self.Out, params))

    p = params.modulus
    cc = compile_circuit(Chain())
    x = 0x1234567890abcdef1234567890abcdef1234567890abcdef12345678 % p
    acc = x
    for i in range(130):
        acc = (2 * acc - x) % p if i % 3 else (acc + 3 * acc) % p
    vec = cc.assignment_vector({"Out": em.ValueOf(acc * acc % p, params), "X": em.ValueOf(x, params)})
    w, *_ = cc.run_vprogram(vec)
    assert cc.last_status == 0 and cc.is_satisfied(w)[0]
    vec[0] ^= 1
    cc.run_vprogram(vec)
    assert cc.last_status != 0


def test_emulated_poseidon_matches_native():
    """TestEmulatedPoseidonMatchesNative (poseidon_test.go:45-78): inputs 1, 2, 3.  The C oracle's
    solver computes the commitment (Pedersen MSM over ~10^5 limbs) and its challenge; the Python
    interpreter is given that challenge (its own Python MSM is checked on the small circuits above
    and in test_commitment.py) and must reproduce every wire and row."""
    cc = H.compiled("emulated-poseidon")
    HashCircuit = circuits.EmulatedPoseidonCircuit
    rh = cref.R1csHandle(cc)
    pk, vk, _ = groth16.setup(cc, 5, _mul)
    ch = cref.CommitKeysHandle(pk)
    vec = cc.assignment_vector(HashCircuit.assignment((1, 2, 3)))
    rc, ow, oa, ob, oc, _ = cref.r1cs_solve_ex(rh, ch, to_mont_array(vec))
    assert rc == 0
    ow = from_mont_array(ow)
    cc.commit_fn = lambda idx, hashed, committed: ow[cc.commitments[idx]["wire"]]
    try:
        w, a, b, c = cc.run_vprogram(vec)
        assert cc.last_status == 0 and cc.is_satisfied(w)[0]
        assert ow == w and from_mont_array(oa) == a and from_mont_array(ob) == b
        bad = cc.assignment_vector(HashCircuit.assignment((1, 2, 3), poseidon_native.hash([1, 2, 4])))
        cc.run_vprogram(bad)
        assert cc.last_status != 0
    finally:
        cc.commit_fn = None
    rc, *_ = cref.r1cs_solve_ex(rh, ch, to_mont_array(bad))
    assert rc != 0


# ---- ecc/format/twistededwards_test.go: native and emulated coordinate changes in one circuit -------
FORMAT_X = 20284931487578954787250358776722960153090567235942462656834196519767860852891
FORMAT_Y = 21185575020764391300398134415668786804224896114060668011215204645513129497221


class FormatCircuit:
    """testFromTwistedEdwards / testToTwistedEdwards (twistededwards_test.go:20-68)"""
    X = Secret()
    Y = Secret()
    XPrime = Secret()
    YPrime = Secret()
    EX = Secret(4)
    EY = Secret(4)
    EXPrime = Secret(4)
    EYPrime = Secret(4)

    def __init__(self, to_te):
        self.to_te = to_te

    def define(self, api):
        from gnark_crypto_primitives_amd.ecc import format as fmt
        sf = em.BN254Fr
        native, emulated_ = (fmt.FromRTEtoTE, fmt.FromEmulatedRTEtoTE) if self.to_te else \
            (fmt.FromTEtoRTE, fmt.FromEmulatedTEtoRTE)
        xp, yp = native(api, self.X, self.Y)
        api.AssertIsEqual(xp, self.XPrime)
        api.AssertIsEqual(yp, self.YPrime)
        exp, eyp = emulated_(api, em.Element(self.EX, sf), em.Element(self.EY, sf))
        field = em.NewField(api, sf)
        field.AssertIsEqual(exp, em.Element(self.EXPrime, sf))
        field.AssertIsEqual(eyp, em.Element(self.EYPrime, sf))


def format_assignment(to_te, x=FORMAT_X, y=FORMAT_Y, bad=False):
    from gnark_crypto_primitives_amd.ecc import format as fmt
    xr, yr = fmt.te_to_rte_native(x, y)
    src, dst = ((xr, yr), (x, y)) if to_te else ((x, y), (xr, yr))
    v = lambda n: em.ValueOf(n, em.BN254Fr)
    return {"X": src[0], "Y": src[1], "XPrime": dst[0], "YPrime": dst[1], "EX": v(src[0]),
            "EY": v(src[1]), "EXPrime": v(dst[0] + (1 if bad else 0)), "EYPrime": v(dst[1])}


@pytest.mark.parametrize("to_te", [False, True], ids=["TEtoRTE", "RTEtoTE"])
def test_format_native_and_emulated(to_te):
    """TestFromTwistedEdwards / TestFromReducedTwistedEdwards (twistededwards_test.go:70-134): the
    reference's point, converted natively and over the emulated field; the limb constants of
    twistededwards.go:18-23 are -f and its inverse."""
    from gnark_crypto_primitives_amd.ecc import format as fmt
    limbs = lambda ls: sum(int(v) << (64 * i) for i, v in enumerate(ls))
    assert limbs(fmt.EMULATED_NEG_SCALING_FACTOR) == -fmt.SCALING_FACTOR % fmt.R
    assert limbs(fmt.EMULATED_INV_NEG_SCALING_FACTOR) == pow(-fmt.SCALING_FACTOR, -1, fmt.R)
    cc = compile_circuit(FormatCircuit(to_te))
    vec = cc.assignment_vector(format_assignment(to_te))
    w, *_ = cc.run_vprogram(vec)
    assert cc.last_status == 0 and cc.is_satisfied(w)[0]
    cc.run_vprogram(cc.assignment_vector(format_assignment(to_te, bad=True)))
    assert cc.last_status != 0
    rh = cref.R1csHandle(cc)
    pk, vk, _ = groth16.setup(cc, 77, _mul)
    ch = cref.CommitKeysHandle(pk)
    rc, *_ = cref.r1cs_solve_ex(rh, ch, to_mont_array(vec))
    assert rc == 0
