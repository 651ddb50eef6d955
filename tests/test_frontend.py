"""R1CS builder semantics (constraint costs, solve order) and solver parity:
witness program (what the GPU runs) == gnark-style constraint solver of the C oracle."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd.frontend import Public, Secret, compile_circuit
from gnark_crypto_primitives_amd.frontend.api import API, CompileError
from gnark_crypto_primitives_amd.frontend.compile import from_mont_array, to_mont_array
from oracle import cref

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _count(fn, n_in=4):
    api = API()
    xs = [api.secret_input(f"x{i}") for i in range(n_in)]
    fn(api, *xs)
    return api.NbConstraints()


def test_constraint_costs():
    assert _count(lambda api, a, b, c, d: api.Add(a, b, 3, c)) == 0
    assert _count(lambda api, a, b, c, d: api.Mul(a, 7)) == 0
    assert _count(lambda api, a, b, c, d: api.Mul(a, b)) == 1
    assert _count(lambda api, a, b, c, d: api.Mul(a, b, c)) == 2
    assert _count(lambda api, a, b, c, d: api.IsZero(a)) == 2
    assert _count(lambda api, a, b, c, d: api.Inverse(a)) == 1
    assert _count(lambda api, a, b, c, d: api.DivUnchecked(a, b)) == 1
    assert _count(lambda api, a, b, c, d: api.AssertIsEqual(a, b)) == 1
    assert _count(lambda api, a, b, c, d: api.AssertIsBoolean(a)) == 1
    assert _count(lambda api, a, b, c, d: api.ToBinary(a, 10)) == 11
    # Select: booleanity of an unknown selector + 1
    assert _count(lambda api, a, b, c, d: api.Select(a, b, c)) == 2
    assert _count(lambda api, a, b, c, d: api.Select(api.IsZero(a), b, c)) == 3
    # And of two already-boolean values: 1
    assert _count(lambda api, a, b, c, d: api.And(api.IsZero(a), api.IsZero(b))) == 5
    # Lookup2 on ToBinary bits: 3 with variable table, 1 with constant table
    assert _count(lambda api, a, b, c, d: api.Lookup2(*api.ToBinary(a, 2), b, c, d, a)) == 3 + 3
    assert _count(lambda api, a, b, c, d: api.Lookup2(*api.ToBinary(a, 2), 5, 6, 7, 9)) == 3 + 1
    # constants fold completely
    assert _count(lambda api, a, b, c, d: api.AssertIsEqual(api.Mul(3, 4), 12)) == 0
    with pytest.raises(CompileError):
        _count(lambda api, a, b, c, d: api.AssertIsEqual(api.Mul(3, 4), 13))


class Mixed:
    """Touches every opcode of the witness program."""
    X = Secret()
    Y = Secret()
    Z = Public()

    def define(self, api):
        bits = api.ToBinary(self.X, 16)
        s = api.FromBinary(*bits[:8])
        iz = api.IsZero(api.Sub(self.X, self.Y))
        q = api.DivUnchecked(self.X, api.Add(self.Y, 1))
        inv = api.Inverse(api.Add(self.Y, 2))
        sel = api.Select(bits[0], q, inv)
        lk = api.Lookup2(bits[1], bits[2], 10, self.Y, s, q)
        x = api.Xor(bits[3], bits[4])
        o = api.Or(bits[5], iz)
        acc = api.Add(api.Mul(sel, lk), api.Neg(x), o, api.Mul(s, s, s))
        api.AssertIsEqual(acc, self.Z)


def _mixed_expected(x, y):
    b = [(x >> i) & 1 for i in range(16)]
    s = x & 0xff
    iz = int(x == y)
    q = x * pow(y + 1, R - 2, R) % R
    inv = pow(y + 2, R - 2, R)
    sel = q if b[0] else inv
    lk = [10, y, s, q][b[1] + 2 * b[2]]
    return (sel * lk - (b[3] ^ b[4]) + (b[5] | iz) + s * s * s) % R


def test_program_vs_c_oracle_solver():
    cc = compile_circuit(Mixed())
    rh = cref.R1csHandle(cc)
    rng = random.Random(1)
    for trial in range(20):
        x = rng.randrange(1 << 16)
        y = x if trial % 5 == 0 else rng.randrange(R)
        inp = cc.assignment_vector({"X": x, "Y": y, "Z": _mixed_expected(x, y)})
        wires, a, b, c = cc.run_program(inp)
        assert cc.last_status == 0 and cc.is_satisfied(wires)[0]
        rc, w2, a2, b2, c2 = cref.r1cs_solve(rh, to_mont_array(inp))
        assert rc == 0
        assert from_mont_array(w2) == wires
        assert (from_mont_array(a2), from_mont_array(b2), from_mont_array(c2)) == (a, b, c)


def test_unsatisfied_detected_by_both_solvers():
    cc = compile_circuit(Mixed())
    rh = cref.R1csHandle(cc)
    x, y = 1234, 99
    bad = cc.assignment_vector({"X": x, "Y": y, "Z": (_mixed_expected(x, y) + 1) % R})
    cc.run_program(bad)
    assert cc.last_status == -5
    assert cref.r1cs_solve(rh, to_mont_array(bad))[0] < 0
    # X does not fit 16 bits: the NBits hint cannot satisfy the recomposition
    big = cc.assignment_vector({"X": 1 << 20, "Y": y, "Z": 0})
    cc.run_program(big)
    assert cc.last_status == -5
    assert cref.r1cs_solve(rh, to_mont_array(big))[0] < 0
    # division by zero: Y + 2 == 0
    z = cc.assignment_vector({"X": x, "Y": R - 2, "Z": 0})
    cc.run_program(z)
    assert cc.last_status == -5
    assert cref.r1cs_solve(rh, to_mont_array(z))[0] < 0


def test_wire_order_and_layout():
    cc = compile_circuit(Mixed())
    assert [n for n, _, _ in cc.layout] == ["Z", "X", "Y"]     # public first (gnark wire order)
    assert cc.n_public == 2 and cc.n_secret == 2
    assert cc.program.shape[1] == 4 and cc.program[-1, 0] == 0
    assert cc.n_slots >= cc.n_wires


def test_to_binary_254_rejects_the_unreduced_decomposition():
    """ADVICE r1 / gnark std/math/bits ToBinary with NbDigits == FieldBitLen [UPSTREAM-RECALL]:
    besides booleanity and recomposition the bits must be <= r - 1.  For s < 2^254 - r the bits of
    s + r recompose to s mod r as well; that assignment satisfies every other row and must fail."""
    from gnark_crypto_primitives_amd.frontend import Public

    class Bits:
        X = Public()
        B0 = Public()

        def define(self, api):
            bits = api.ToBinary(self.X, 254)
            api.AssertIsEqual(bits[0], self.B0)

    cc = compile_circuit(Bits())
    # 254 booleanity rows + recomposition + equality + MustBeLessOrEqCst(r - 1)
    ones = bin(R - 1).count("1")
    assert cc.n_constraints == 254 + 1 + 1 + (ones - 1) + (254 - ones)
    s = 12345
    wires, *_ = cc.run_program(cc.assignment_vector({"X": s, "B0": s & 1}))
    assert cc.last_status == 0 and cc.is_satisfied(wires)[0]
    # forge: replace the bit wires by the bits of s + r (still 254 bits) and re-derive the
    # products p[i] the comparison allocates, so that only the zero-bit rows can object
    assert (s + R) >> 254 == 0
    alias = [((s + R) >> i) & 1 for i in range(254)]
    bit_wires = list(range(cc.n_public, cc.n_public + 254))
    forged = list(wires)
    for w, b in zip(bit_wires, alias):
        forged[w] = b
    p = 1
    nxt = cc.n_public + 254
    first = True
    for i in range(253, -1, -1):
        if ((R - 1) >> i) & 1:
            p = p * alias[i]
            if first:
                first = False          # p[253] = 1 * a[253]: no wire
            else:
                forged[nxt] = p
                nxt += 1
    assert nxt == cc.n_wires
    forged[2] = alias[0]               # public B0 follows the forged bit 0
    ok, row = cc.is_satisfied(forged)
    assert not ok
    L, Rr, O, *_ = cc.constraints[row]
    assert not O                        # the failing row is a (1 - p - a) * a == 0 comparison row
    # AssertIsLessOrEqual against a small constant
    class Le:
        X = Public()

        def define(self, api):
            api.AssertIsLessOrEqual(self.X, 1000)

    cc = compile_circuit(Le())
    for x, good in ((0, True), (1000, True), (1001, False), (R - 1, False)):
        cc.run_program(cc.assignment_vector({"X": x}))
        assert (cc.last_status == 0) == good, x


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64, 0])
def test_vliw_schedule_equals_the_sequential_program(lanes):
    """The scheduled program the GPU runs (vprogram: steps of up to S independent operations, slots
    recycled by step) computes the same wires, rows and status as the sequential program, on a
    circuit that touches every opcode (incl. the hoisted batch inversion) and on an SMT verifier;
    every constraint row is emitted exactly once."""
    from gnark_crypto_primitives_amd import circuits
    from gnark_crypto_primitives_amd.tree import smt_witness
    rng = random.Random(lanes)
    cases = []
    cc = compile_circuit(Mixed(), lanes)
    for i in range(6):
        x = rng.randrange(1 << 16)
        y = x if i % 3 == 0 else rng.randrange(R)
        cases.append((cc, cc.assignment_vector({"X": x, "Y": y, "Z": _mixed_expected(x, y)}), 0))
    cases.append((cc, cc.assignment_vector({"X": 7, "Y": 9, "Z": 1}), -5))
    cases.append((cc, cc.assignment_vector({"X": 1 << 20, "Y": 9, "Z": 0}), -5))
    cc2 = compile_circuit(circuits.smt_inclusion_circuit(12), lanes)
    for k in (0, 3, 11):
        w = smt_witness.synthetic_inclusion(rng, 12, k)
        cases.append((cc2, cc2.assignment_vector(w), 0))
    w["Root"] = (w["Root"] + 1) % R
    cases.append((cc2, cc2.assignment_vector(w), -5))
    for c, inp, want_status in cases:
        assert c.lanes_per_proof in (1, 2, 4, 8, 16, 32, 64)
        if lanes:
            assert c.lanes_per_proof == lanes
        seq = c.run_program(inp)
        st_seq = c.last_status
        par = c.run_vprogram(inp)
        assert c.last_status == st_seq == want_status
        assert par[0] == seq[0]                       # wires
        assert par[1:] == seq[1:]                     # a, b, c rows (none missing)
        # header sanity: active counts, sub-lane capacity
        S = c.lanes_per_proof
        assert c.vprogram.shape[1:] == (1 + S, 4)
    # more sub-lanes never lengthen the schedule
    if lanes in (2, 4):
        assert compile_circuit(circuits.smt_inclusion_circuit(12), lanes).v_n_steps < \
            compile_circuit(circuits.smt_inclusion_circuit(12), 1).v_n_steps
