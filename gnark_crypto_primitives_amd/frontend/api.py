"""R1CS builder exposing the subset of gnark's ``frontend.API`` that the reference gadgets call.

The reference's gadgets are Go functions over ``frontend.API`` (e.g.
hash/native/bn254/poseidon/poseidon.go:154-232, tree/smt/*.go, elgamal/mul.go:86-161); the
builder behind that interface lives in the third-party module gnark (go.mod:8, not in
/root/reference).  This module restates the behaviour of gnark's ``frontend/cs/r1cs`` builder
that those call sites rely on [UPSTREAM-RECALL, SURVEY.md §8a]:

* a ``Variable`` is a linear expression over wires; wire 0 is the constant ONE;
* Add/Sub/Neg and multiplication by a constant fold into the expression and cost nothing;
* ``Mul`` of two non-constant expressions allocates one internal wire and one constraint;
* ``IsZero`` = InvZero hint + 2 constraints; ``ToBinary(n)`` = NBits hint + n booleanity
  constraints + 1 equality; ``Select``/``And``/``Or``/``Xor`` = 1 constraint (+1 per operand not
  yet known to be boolean); ``Lookup2`` = 2 constraints; ``AssertIsEqual`` = 1 constraint.

Besides the R1CS (L, R, O in CSR form plus the solve order), the builder records a
straight-line *witness program* in SSA form: every API call appends the field operation that
produces its value.  The GPU solver (csrc/solve.hip) runs that program with one lane per proof;
the per-constraint evaluations a_k = <L_k,w>, b_k, c_k that Groth16's quotient needs are then
simply the operand values of the constraint-producing instructions.
"""
from __future__ import annotations

import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

# witness-program opcodes (decoded by csrc/solve.hip and by CompiledCircuit.run_program)
OP_END, OP_ADD, OP_SUB, OP_MUL, OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_BITS, OP_SETC, OP_ABC, \
    OP_COPY, OP_DIV, OP_BATCHINV, OP_PAIR, OP_MULABC, OP_XORABC, OP_XOR = range(18)
# OP_XOR: OP_XORABC without its R1CS row (the PLONK lowering emits its own gate rows, scs.py)
# 18, 19 = OP_FMAC, OP_FMA (relin.py).  Units with operand rows (like OP_BATCHINV + OP_PAIR):
#   (OP_HIST, first value, n queries, table size) + n x (OP_HQ, 0, query value, 0): multiplicities
#     m_j = #{queries equal to j}, j < table size, into the consecutive wires first .. first + size - 1
#   (OP_COMMIT, challenge value, n operands, commitment index) + n x (OP_HQ, 0, value, 0): the
#     solver stops here; the host commits to the operand wires (Pedersen MSM), hashes and writes the
#     challenge into the value's slot, then the program continues
# OP_BITS with b = count | width << 16: `count` limbs of `width` bits each (width 0 / 1: bits).
OP_HIST, OP_HQ, OP_COMMIT = 20, 21, 22
# bitwise XOR / AND of two small integers (< 2^32): the result hints of gnark's byte lookup tables
# (std/internal/logderivprecomp: uints.Xor / And); unconstrained, the lookup constrains them
OP_BXOR, OP_BAND = 23, 24
# quotient and remainder of a product of two multi-limb integers by a 4 x 64-bit modulus (the mulHint
# of gnark's std/math/emulated): a unit like OP_HIST,
#   (OP_EMUL, first value, na + nb, nout | na << 8 | first const of the modulus << 12)
#   + na + nb x (OP_HQ, 0, limb value, 0): the limbs of a, then of b, least significant first
# a = sum a_i 2^(64 i), b likewise, p = sum const[first + i] 2^(64 i), i < 4: the nout - 4 64-bit limbs
# of floor(a b / p), then the four of a b mod p, into consecutive wires
OP_EMUL = 25
EMUL_LIMB_BITS, EMUL_LIMBS = 64, 4

HINT_INVZERO, HINT_NBITS, HINT_LIMBS, HINT_COUNT, HINT_COMMIT, HINT_BYTEOP, HINT_EMUL = \
    1, 2, 3, 4, 5, 6, 7
FIELD_BITS = 254      # bit length of r


class Variable:
    """Linear expression sum_i coef_i * wire_i (wire 0 = ONE) plus the SSA value that holds it."""
    __slots__ = ("lc", "val")

    def __init__(self, lc, val):
        self.lc = lc      # dict wire -> coef (non-zero, reduced mod R)
        self.val = val    # SSA value id

    def is_const(self):
        return all(w == 0 for w in self.lc)

    def const_value(self):
        return self.lc.get(0, 0)


def _lc_add(a, b, sb=1):
    out = dict(a)
    for w, c in b.items():
        v = (out.get(w, 0) + sb * c) % R
        if v:
            out[w] = v
        else:
            out.pop(w, None)
    return out


def _lc_scale(a, k):
    k %= R
    if k == 0:
        return {}
    return {w: c * k % R for w, c in a.items()}


class CompileError(Exception):
    pass


class API:
    """Mirror of the calls the reference gadgets make on gnark's frontend.API."""

    def __init__(self):
        self.n_wires = 1            # wire 0 = ONE
        self.n_public = 1
        self.n_secret = 0
        self.constraints = []       # (L, R, O, solve_wire, (va, vb, vc), check)
        self.instr = []             # (kind, index): 0 = R1C, 1 = hint
        self.hints = []             # (kind, [input lc], [output wires])
        self.ops = []               # SSA ops: (opcode, dst_val, a, b) ; see _emit
        self.n_vals = 0
        self.val_wire = {}          # SSA value id -> wire whose slot it lives in
        self.consts = {}            # int -> const pool index
        self.const_list = []
        self.booleans = set()
        self._inputs_open = True
        self.input_names = []
        self.val_one = self._new_val()
        self.val_wire[self.val_one] = 0
        self.val_zero = None
        self.println_log = []
        self.commitments = []       # {"private": [wires], "hashed": [wires], "wire": w}
        self._deferred = []
        self._emul_moduli = {}      # modulus -> index of its first limb in the constant pool

    # ------------------------------------------------------------------ plumbing
    def _new_val(self):
        v = self.n_vals
        self.n_vals += 1
        return v

    def _cid(self, c):
        c %= R
        i = self.consts.get(c)
        if i is None:
            i = len(self.const_list)
            self.consts[c] = i
            self.const_list.append(c)
        return i

    def _emit(self, op, a=0, b=0, dst=None):
        if dst is None:
            dst = self._new_val()
        self.ops.append((op, dst, a, b))
        return dst

    def _new_wire(self, val=None):
        w = self.n_wires
        self.n_wires += 1
        if val is not None:
            self.val_wire[val] = w
        return w

    def _input(self, name, public):
        if not self._inputs_open:
            raise CompileError("inputs must be declared before any constraint")
        if public and self.n_secret:
            raise CompileError("public inputs come first (gnark wire order)")
        v = self._new_val()
        w = self._new_wire(v)
        if public:
            self.n_public += 1
        else:
            self.n_secret += 1
        self.input_names.append(name)
        return Variable({w: 1}, v)

    def public_input(self, name):
        return self._input(name, True)

    def secret_input(self, name):
        return self._input(name, False)

    def _const(self, c):
        c %= R
        if c == 1:
            return Variable({0: 1}, self.val_one)
        v = self._emit(OP_SETC, 0, self._cid(c))
        return Variable({0: c} if c else {}, v)

    def _v(self, x):
        if isinstance(x, Variable):
            return x
        if isinstance(x, (int, np.integer)):
            return self._const(int(x))
        raise CompileError(f"unsupported operand {type(x)}")

    @staticmethod
    def _key(v):
        return tuple(sorted(v.lc.items()))

    def _add_r1c(self, L, R_, O, solve_wire=-1, check=None, fused=None):
        """check: the solver must verify L*R == O (assertions; divisions, where a zero divisor
        makes the defining equation unsatisfiable).
        fused: (opcode, dst, a, b) -- one witness-program op that both computes the new wire and
        emits the row (OP_MULABC: dst = a*b, row (a, b, dst); OP_XORABC: dst = a xor b, row
        (2a, b, 2ab)) instead of a value op followed by OP_ABC."""
        self._inputs_open = False
        k = len(self.constraints)
        if check is None:
            check = solve_wire < 0
        self.constraints.append((L.lc, R_.lc, O.lc, solve_wire, (L.val, R_.val, O.val), check))
        self.instr.append((0, k))
        if fused is not None:
            self.ops.append(fused)
        else:
            self._emit(OP_ABC, R_.val, O.val, dst=L.val)   # (sa, sb, sc) = (dst, a, b)
        return k

    def _internal(self, val):
        """Fresh internal wire whose value is SSA value `val`."""
        w = self._new_wire(val)
        return Variable({w: 1}, val), w

    # ------------------------------------------------------------------ arithmetic
    def Add(self, a, b, *rest):
        res = self._v(a)
        for x in (b,) + rest:
            x = self._v(x)
            if not x.lc:
                continue
            if not res.lc:
                res = x
                continue
            if x.is_const():
                val = self._emit(OP_ADDC, res.val, self._cid(x.const_value()))
            elif res.is_const():
                val = self._emit(OP_ADDC, x.val, self._cid(res.const_value()))
            else:
                val = self._emit(OP_ADD, res.val, x.val)
            res = Variable(_lc_add(res.lc, x.lc), val)
        return res

    def Sum(self, xs):
        """Add over a long list: the linear expression is merged in place (a chain of Add would
        copy the growing expression every time) and the values are added pairwise (a tree, so the
        witness program's dependency chain is logarithmic)."""
        xs = [self._v(x) for x in xs]
        xs = [x for x in xs if x.lc]
        if not xs:
            return self._const(0)
        lc = {}
        for x in xs:
            for w, c in x.lc.items():
                v = (lc.get(w, 0) + c) % R
                if v:
                    lc[w] = v
                else:
                    lc.pop(w, None)
        vals = [x.val for x in xs]
        while len(vals) > 1:
            nxt = [self._emit(OP_ADD, vals[i], vals[i + 1]) for i in range(0, len(vals) - 1, 2)]
            if len(vals) % 2:
                nxt.append(vals[-1])
            vals = nxt
        return Variable(lc, vals[0])

    def Neg(self, a):
        a = self._v(a)
        if not a.lc:
            return a
        if a.is_const():
            return self._const(-a.const_value())
        return Variable(_lc_scale(a.lc, R - 1), self._emit(OP_NEG, a.val))

    def Sub(self, a, b, *rest):
        res = self._v(a)
        for x in (b,) + rest:
            x = self._v(x)
            if not x.lc:
                continue
            if not res.lc:
                res = self.Neg(x)
                continue
            if x.is_const():
                val = self._emit(OP_ADDC, res.val, self._cid(-x.const_value()))
            elif res.is_const():
                val = self._emit(OP_ADDC, self.Neg(x).val, self._cid(res.const_value()))
            else:
                val = self._emit(OP_SUB, res.val, x.val)
            res = Variable(_lc_add(res.lc, x.lc, -1), val)
        return res

    def _mul2(self, a, b):
        a, b = self._v(a), self._v(b)
        if not a.lc or not b.lc:
            return self._const(0)
        if a.is_const() and b.is_const():
            return self._const(a.const_value() * b.const_value())
        if a.is_const() or b.is_const():
            k, x = (a, b) if a.is_const() else (b, a)
            kv = k.const_value()
            if kv == 1:
                return x
            return Variable(_lc_scale(x.lc, kv), self._emit(OP_MULC, x.val, self._cid(kv)))
        val = self._new_val()
        res, w = self._internal(val)
        self._add_r1c(a, b, res, solve_wire=w, fused=(OP_MULABC, val, a.val, b.val))
        return res

    def Mul(self, a, b, *rest):
        res = self._mul2(a, b)
        for x in rest:
            res = self._mul2(res, x)
        return res

    def MulAcc(self, a, b, c):
        return self.Add(a, self.Mul(b, c))

    def Inverse(self, a):
        """res * a == 1 (gnark r1cs builder: one constraint, solved by division)."""
        a = self._v(a)
        if a.is_const():
            if a.const_value() == 0:
                raise CompileError("inverse of constant zero")
            return self._const(pow(a.const_value(), R - 2, R))
        val = self._emit(OP_INV, a.val)
        res, w = self._internal(val)
        one = self._const(1)
        self._add_r1c(res, a, one, solve_wire=w, check=True)
        return res

    def DivUnchecked(self, a, b):
        """res * b == a (0/0 = 0 as in gnark's solver)."""
        a, b = self._v(a), self._v(b)
        if b.is_const():
            if b.const_value() == 0:
                raise CompileError("division by constant zero")
            return self._mul2(a, pow(b.const_value(), R - 2, R))
        if not a.lc:
            return self._const(0)
        val = self._emit(OP_DIV, a.val, b.val)
        res, w = self._internal(val)
        self._add_r1c(res, b, a, solve_wire=w, check=True)
        return res

    def _batch_inverse_vals(self, xs):
        """SSA values of 1 / x_i (0 for x_i = 0) from ONE field inversion: an OP_BATCHINV unit
        (Montgomery's trick in the solver kernel).  Values only: no wires, no constraints."""
        vals = [self._new_val() for _ in xs]
        self.ops.append((OP_BATCHINV, len(xs), 0, 0))
        for v, x in zip(vals, xs):
            self.ops.append((OP_PAIR, v, x.val, 0))
        return vals

    def InverseBatch(self, xs):
        """[Inverse(x) for x in xs] -- the same wires and constraints (res_i * x_i == 1) -- with the
        witness computed by one shared inversion instead of one per element (the lookup arguments
        invert tens of thousands of independent values)."""
        xs = [self._v(x) for x in xs]
        if any(x.is_const() for x in xs) or len(xs) < 4:
            return [self.Inverse(x) for x in xs]
        out = []
        one = self._const(1)
        for val, x in zip(self._batch_inverse_vals(xs), xs):
            res, w = self._internal(val)
            self._add_r1c(res, x, one, solve_wire=w, check=True)
            out.append(res)
        return out

    def DivUncheckedBatch(self, nums, dens):
        """[DivUnchecked(a, b) for a, b in zip(nums, dens)] with one shared inversion (0 / 0 = 0)."""
        nums, dens = [self._v(a) for a in nums], [self._v(b) for b in dens]
        if any(b.is_const() for b in dens) or any(not a.lc for a in nums) or len(dens) < 4:
            return [self.DivUnchecked(a, b) for a, b in zip(nums, dens)]
        out = []
        for inv, a, b in zip(self._batch_inverse_vals(dens), nums, dens):
            val = self._emit(OP_MUL, a.val, inv)
            res, w = self._internal(val)
            self._add_r1c(res, b, a, solve_wire=w, check=True)
            out.append(res)
        return out

    def Div(self, a, b):
        """gnark Div: asserts b != 0 by computing its inverse, then multiplies."""
        a, b = self._v(a), self._v(b)
        if b.is_const():
            return self.DivUnchecked(a, b)
        return self._mul2(a, self.Inverse(b))

    # ------------------------------------------------------------------ booleans
    def _mark_boolean(self, v):
        self.booleans.add(self._key(v))

    def _is_boolean(self, v):
        if v.is_const():
            return v.const_value() in (0, 1)
        return self._key(v) in self.booleans

    def AssertIsBoolean(self, a):
        a = self._v(a)
        if a.is_const():
            if a.const_value() not in (0, 1):
                raise CompileError("constant is not boolean")
            return
        if self._is_boolean(a):
            return
        self._mark_boolean(a)
        # a * (1 - a) == 0
        self._add_r1c(a, self.Sub(1, a), self._zero())

    def _zero(self):
        if self.val_zero is None:
            self.val_zero = self._emit(OP_SETC, 0, self._cid(0))
        return Variable({}, self.val_zero)

    def IsZero(self, a):
        a = self._v(a)
        if a.is_const():
            return self._const(1 if a.const_value() == 0 else 0)
        # x = 1/a (0 if a == 0) from the InvZero hint; m = 1 - a*x ; a*m == 0
        xval = self._emit(OP_INV, a.val)
        x, xw = self._internal(xval)
        self.hints.append((HINT_INVZERO, [a.lc], [xw]))
        self.instr.append((1, len(self.hints) - 1))
        nega = self.Neg(a)
        cval = self._emit(OP_MUL, nega.val, x.val)          # = m - 1
        mval = self._emit(OP_ADDC, cval, self._cid(1))
        m, mw = self._internal(mval)
        mm1 = Variable(_lc_add(m.lc, {0: 1}, -1), cval)
        self._add_r1c(nega, x, mm1, solve_wire=mw)
        self._add_r1c(a, m, self._zero())
        self._mark_boolean(m)
        return m

    def Select(self, cond, a, b):
        cond, a, b = self._v(cond), self._v(a), self._v(b)
        self.AssertIsBoolean(cond)
        if cond.is_const():
            return a if cond.const_value() == 1 else b
        if a.is_const() and b.is_const():
            return self.Add(self._mul2(cond, (a.const_value() - b.const_value()) % R), b)
        return self.Add(self._mul2(cond, self.Sub(a, b)), b)

    def Lookup2(self, b0, b1, i0, i1, i2, i3):
        """gnark r1cs Lookup2: two constraints."""
        b0, b1 = self._v(b0), self._v(b1)
        i0, i1, i2, i3 = (self._v(x) for x in (i0, i1, i2, i3))
        self.AssertIsBoolean(b0)
        self.AssertIsBoolean(b1)
        if b0.is_const() and b1.is_const():
            return (i0, i1, i2, i3)[b0.const_value() + 2 * b1.const_value()]
        # tmp = b1 * (i3 - i2 - i1 + i0); res = (tmp + i1 - i0) * b0 + (i2 - i0) * b1 + i0
        tmp = self.Add(self.Sub(i3, i2, i1), i0)
        tmp1 = self._mul2(tmp, b1)
        tmp1 = self.Sub(self.Add(tmp1, i1), i0)
        tmp2 = self._mul2(tmp1, b0)
        tmp3 = self._mul2(self.Sub(i2, i0), b1)
        return self.Add(tmp2, tmp3, i0)

    def And(self, a, b):
        a, b = self._v(a), self._v(b)
        self.AssertIsBoolean(a)
        self.AssertIsBoolean(b)
        res = self._mul2(a, b)
        if not res.is_const():
            self._mark_boolean(res)
        return res

    def Or(self, a, b):
        a, b = self._v(a), self._v(b)
        self.AssertIsBoolean(a)
        self.AssertIsBoolean(b)
        # a + b - ab
        res = self.Sub(self.Add(a, b), self._mul2(a, b))
        if not res.is_const():
            self._mark_boolean(res)
        return res

    def Xor(self, a, b):
        """gnark r1cs builder: a fresh wire res with (2a) * b == a + b - res (one constraint, and
        the result is a single wire however long the XOR chain grows)."""
        a, b = self._v(a), self._v(b)
        self.AssertIsBoolean(a)
        self.AssertIsBoolean(b)
        if a.is_const() or b.is_const():
            k, x = (a, b) if a.is_const() else (b, a)
            res = x if k.const_value() == 0 else self.Sub(1, x)
            if not res.is_const():
                self._mark_boolean(res)
            return res
        val = self._new_val()
        res, w = self._internal(val)
        # the row's operand values are produced by the fused op itself (no SSA ids needed)
        two_a = Variable(_lc_scale(a.lc, 2), -1)
        rhs = Variable(_lc_add(_lc_add(a.lc, b.lc), res.lc, -1), -1)
        self._add_r1c(two_a, b, rhs, solve_wire=w, fused=(OP_XORABC, val, a.val, b.val))
        self._mark_boolean(res)
        return res

    # ------------------------------------------------------------------ bits
    def ToBinary(self, a, n=254, omit_modulus_check=False):
        """std/math/bits.ToBinary with WithNbDigits(n): LSB first.  With n = the field's bit
        length gnark also asserts that the bits are <= r - 1 (``omitReducednessCheck`` is false
        unless n < FieldBitLen or OmitModulusCheck() is passed) [UPSTREAM-RECALL]."""
        a = self._v(a)
        if a.is_const():
            v = a.const_value()
            if v >> n:
                raise CompileError("constant does not fit")
            return [self._const((v >> i) & 1) for i in range(n)]
        first = self._new_val()
        vals = [first] + [self._new_val() for _ in range(n - 1)]
        self.ops.append((OP_BITS, first, a.val, n))
        wires = [self._new_wire(v) for v in vals]
        self.hints.append((HINT_NBITS, [a.lc], wires))
        self.instr.append((1, len(self.hints) - 1))
        bits = [Variable({w: 1}, v) for w, v in zip(wires, vals)]
        acc = self._const(0)
        for i, b in enumerate(bits):
            self.AssertIsBoolean(b)
            acc = self.Add(acc, self._mul2(b, pow(2, i, R)))
        self.AssertIsEqual(acc, a)
        if n >= FIELD_BITS and not omit_modulus_check:
            # reducedness: without it any s < 2^254 - r also decomposes as the bits of s + r
            self._must_be_less_or_eq_cst(bits, R - 1)
        return bits

    def _must_be_less_or_eq_cst(self, bits, bound):
        """gnark r1cs builder ``MustBeLessOrEqCst(aBits, bound)`` [UPSTREAM-RECALL]: with
        p[i] = AND of a[j] for the one-bits j >= i of the bound ("a equals the bound down to bit
        i"), a zero-bit of the bound forces a[i] = 0 while the prefix is still equal:
        (1 - p[i+1] - a[i]) * a[i] == 0.  One product per one-bit above the bound's trailing ones,
        one row per zero-bit."""
        nb = len(bits)
        if bound >> nb:
            return                      # every nb-bit value is below the bound
        t = 0
        while t < nb and (bound >> t) & 1:
            t += 1
        p = [None] * (nb + 1)
        p[nb] = self._const(1)
        for i in range(nb - 1, t - 1, -1):
            p[i] = self._mul2(p[i + 1], bits[i]) if (bound >> i) & 1 else p[i + 1]
        for i in range(nb - 1, -1, -1):
            if (bound >> i) & 1:
                self.AssertIsBoolean(bits[i])
            else:
                self._add_r1c(self.Sub(1, p[i + 1], bits[i]), bits[i], self._zero())
                self._mark_boolean(bits[i])

    def AssertIsLessOrEqual(self, v, bound):
        """frontend.API.AssertIsLessOrEqual for a constant bound (the only form the gadgets and
        std/math/bits use): binary decomposition of v, then MustBeLessOrEqCst."""
        b = self._v(bound)
        if not b.is_const():
            raise CompileError("AssertIsLessOrEqual: only constant bounds are supported")
        v = self._v(v)
        if v.is_const():
            if v.const_value() > b.const_value():
                raise CompileError("constant exceeds the bound")
            return
        bits = self.ToBinary(v, FIELD_BITS, omit_modulus_check=True)
        self._must_be_less_or_eq_cst(bits, b.const_value())

    def FromBinary(self, *bits):
        acc = self._const(0)
        for i, b in enumerate(bits):
            b = self._v(b)
            self.AssertIsBoolean(b)
            acc = self.Add(acc, self._mul2(b, pow(2, i, R)))
        return acc

    # ------------------------------------------------------------------ hints and commitments
    def Defer(self, fn):
        """frontend.Compiler.Defer: fn(api) runs after the circuit's Define (the range checker and
        the multi-commitment build their arguments there)."""
        self._deferred.append(fn)

    def finalize(self):
        while self._deferred:
            self._deferred.pop(0)(self)

    def _single_wire(self, v):
        """wire index of a variable that is exactly one wire (coefficient 1); any other expression
        gets a fresh internal wire equal to it (one constraint), as gnark's Commit does."""
        if len(v.lc) == 1:
            (w, c), = v.lc.items()
            if c == 1 and w != 0:
                return w, v
        val = self._emit(OP_COPY, v.val)
        res, w = self._internal(val)
        self._add_r1c(self._const(1), v, res, solve_wire=w)
        return w, res

    def NewHintLimbs(self, a, width, count):
        """Unconstrained hint (std/rangecheck DecomposeHint, uints toBytes): ``count`` limbs of
        ``width`` bits of a, least significant first, as fresh internal wires."""
        a = self._v(a)
        if not 1 <= width <= 16 or not 1 <= count <= 256 or width * count > 256:
            raise CompileError("limb hint: width in [1,16], width * count <= 256")
        first = self._new_val()
        vals = [first] + [self._new_val() for _ in range(count - 1)]
        self.ops.append((OP_BITS, first, a.val, count | width << 16))
        wires = [self._new_wire(v) for v in vals]
        self.hints.append((HINT_LIMBS, [{0: width}, a.lc], wires))
        self.instr.append((1, len(self.hints) - 1))
        return [Variable({w: 1}, v) for w, v in zip(wires, vals)]

    def NewHintByteOp(self, op, x, y):
        """Unconstrained hint (uints xorHint / andHint): x XOR y or x AND y of two small integers,
        as a fresh internal wire.  op: OP_BXOR | OP_BAND."""
        if op not in (OP_BXOR, OP_BAND):
            raise CompileError("byte-op hint: OP_BXOR or OP_BAND")
        x, y = self._v(x), self._v(y)
        if x.is_const() and y.is_const():
            a, b = x.const_value(), y.const_value()
            return self._const(a ^ b if op == OP_BXOR else a & b)
        val = self._emit(op, x.val, y.val)
        res, w = self._internal(val)
        self.hints.append((HINT_BYTEOP, [{0: 1 if op == OP_BXOR else 2}, x.lc, y.lc], [w]))
        self.instr.append((1, len(self.hints) - 1))
        return res

    def NewHintCount(self, queries, table_size):
        """Unconstrained hint (std/lookup/logderivarg countHint for the table 0 .. size - 1):
        multiplicity of every table entry among the queries, as fresh internal wires."""
        qs = [self._v(q) for q in queries]
        first = self._new_val()
        vals = [first] + [self._new_val() for _ in range(table_size - 1)]
        self.ops.append((OP_HIST, first, len(qs), table_size))
        for q in qs:
            self.ops.append((OP_HQ, 0, q.val, 0))
        wires = [self._new_wire(v) for v in vals]
        self.hints.append((HINT_COUNT, [{0: table_size}] + [q.lc for q in qs], wires))
        self.instr.append((1, len(self.hints) - 1))
        return [Variable({w: 1}, v) for w, v in zip(wires, vals)]

    def NewHintEmulMul(self, a_limbs, b_limbs, modulus, n_quotient_limbs):
        """Unconstrained hint (std/math/emulated mulHint [UPSTREAM-RECALL]): for the integers
        a = sum a_i 2^(64 i) and b = sum b_i 2^(64 i) (limbs may exceed 64 bits: lazy additions) and
        the modulus p < 2^256, fresh internal wires holding
            k: the ``n_quotient_limbs`` 64-bit limbs of floor(a b / p),
            r: the four 64-bit limbs of a b mod p,
            c: the carries of  a(X) b(X) - r(X) - k(X) p(X) = (2^64 - X) c(X)  (signed, as field
               elements; max(na + nb, nk + 4) - 2 of them).
        Every limb product must stay below the field size (the caller bounds the overflows).
        Witness side: k and r come from one OP_EMUL unit (big-integer division in the solver); the
        carries are field arithmetic on the limbs, c_j = (t_j - r_j - (k p)_j + c_(j-1)) / 2^64."""
        a_limbs = [self._v(x) for x in a_limbs]
        b_limbs = [self._v(x) for x in b_limbs]
        na, nb, nk, nr = len(a_limbs), len(b_limbs), int(n_quotient_limbs), EMUL_LIMBS
        # 2^224 <= p: the solver's division normalises on the top 32-bit word of p (csrc/emul.h)
        if not (1 <= na <= 4 and 1 <= nb <= 4 and 1 <= nk <= 8) or not 1 << 224 <= modulus < 1 << 256:
            raise CompileError("emulated product hint: 1..4 limbs per operand, 1..8 quotient limbs, "
                               "2^224 <= modulus < 2^256")
        p_limbs = [(modulus >> (64 * i)) & (2**64 - 1) for i in range(nr)]
        # the modulus' limbs sit in consecutive constants (appended together: _cid only shares)
        first_c = self._emul_moduli.get(modulus)
        if first_c is None:
            first_c = self._emul_moduli[modulus] = len(self.const_list)
            for i, pl in enumerate(p_limbs):
                self.const_list.append(pl)
                self.consts.setdefault(pl, first_c + i)
        if first_c + nr > 1 << 20:
            raise CompileError("constant pool too large for an emulated-product unit")
        nout = nk + nr
        first = self._new_val()
        vals = [first] + [self._new_val() for _ in range(nout - 1)]
        self.ops.append((OP_EMUL, first, na + nb, nout | na << 8 | first_c << 12))
        for x in a_limbs + b_limbs:
            self.ops.append((OP_HQ, 0, x.val, 0))
        wires = [self._new_wire(v) for v in vals]
        k = [Variable({w: 1}, v) for w, v in zip(wires[:nk], vals[:nk])]
        r = [Variable({w: 1}, v) for w, v in zip(wires[nk:], vals[nk:])]
        # carries: plain value arithmetic (no constraints)
        ncols = max(na + nb - 1, nk + nr - 1)
        inv64 = self._cid(pow(1 << 64, -1, R))
        kp = [self.Sum([self._mul2(k[i], p_limbs[j - i]) for i in range(nk)
                        if 0 <= j - i < nr and p_limbs[j - i]]) for j in range(ncols)]
        carries, prev = [], None
        for j in range(ncols - 1):
            t = None
            for i in range(na):
                if 0 <= j - i < nb:
                    m = self._emit(OP_MUL, a_limbs[i].val, b_limbs[j - i].val)
                    t = m if t is None else self._emit(OP_ADD, t, m)
            sub = self.Add(r[j], kp[j]) if j < nr else kp[j]
            d = self._emit(OP_SUB, t, sub.val) if t is not None else self._emit(OP_NEG, sub.val)
            if prev is not None:
                d = self._emit(OP_ADD, d, prev)
            prev = self._emit(OP_MULC, d, inv64)
            carries.append(prev)
        c_wires = [self._new_wire(v) for v in carries]
        c = [Variable({w: 1}, v) for w, v in zip(c_wires, carries)]
        self.hints.append((HINT_EMUL, [{0: na | nb << 8 | nk << 16}] + [x.lc for x in a_limbs + b_limbs] +
                           [{0: pl} if pl else {} for pl in p_limbs], wires + c_wires))
        self.instr.append((1, len(self.hints) - 1))
        return k, r, c

    def Commit(self, *vs):
        """frontend.Committer.Commit (gnark r1cs builder, Groth16 commitment extension
        [UPSTREAM-RECALL, SURVEY.md §3.2 step 6]): returns a wire holding
        H(Pedersen commitment to the private operands || public operands).  Constants are dropped;
        public wires and earlier commitment wires are hashed, every other operand is a basis point
        of the commitment key.  The solver cannot produce the value itself: the program stops at an
        OP_COMMIT unit and the prover supplies it (csrc/commit.hip)."""
        self._inputs_open = False
        priv, hashed, seen, reads = [], [], set(), []
        cwires = {c["wire"] for c in self.commitments}
        for x in vs:
            x = self._v(x)
            if x.is_const():
                continue
            w, x = self._single_wire(x)
            if w in seen:
                continue
            seen.add(w)
            (hashed if (w < self.n_public or w in cwires) else priv).append(w)
            reads.append(x.val)
        if not priv and not hashed:
            raise CompileError("Commit: nothing to commit to")
        val = self._new_val()
        res, w = self._internal(val)
        idx = len(self.commitments)
        self.ops.append((OP_COMMIT, val, len(reads), idx))
        for r in reads:
            self.ops.append((OP_HQ, 0, r, 0))
        self.commitments.append({"private": priv, "hashed": hashed, "wire": w})
        self.hints.append((HINT_COMMIT, [{0: len(hashed)}] + [{x: 1} for x in hashed + priv], [w]))
        self.instr.append((1, len(self.hints) - 1))
        return res

    # ------------------------------------------------------------------ assertions
    def AssertIsEqual(self, a, b):
        a, b = self._v(a), self._v(b)
        if a.is_const() and b.is_const():
            if a.const_value() != b.const_value():
                raise CompileError("constants differ")
            return
        # encoded 1 * a == b
        self._add_r1c(self._const(1), a, b)

    def AssertIsDifferent(self, a, b):
        self.Inverse(self.Sub(a, b))

    def Println(self, *args):
        self.println_log.append(args)

    def NbConstraints(self):
        return len(self.constraints)
