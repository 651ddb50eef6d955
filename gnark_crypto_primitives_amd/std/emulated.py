"""The slice of gnark's ``std/math/emulated`` the reference's byte helpers touch: an element of a
foreign field as little-endian limbs (``Element.Limbs``), ``ValueOf`` for assignments and the two
parameter sets the reference instantiates (Secp256k1Fp/Fr: 4 limbs x 64 bits).  The
arithmetic (``Field``: Add, Sub, Mul, Reduce, AssertIsEqual, IsZero, Select, ToBits) follows below:
the reference uses it in hash/emulated/bn254/poseidon and tree/smt/emulated."""


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


BN254Fr = FieldParams(
    "BN254Fr", 21888242871839275222246405745257275088548364400416034343698204186575808495617)
BN254Fp = FieldParams(
    "BN254Fp", 21888242871839275222246405745257275088696311157297823662689037894645226208583)


# ---- emulated arithmetic: gnark std/math/emulated Field[T] [UPSTREAM-RECALL, parity unpinned] -------
#
# What the reference calls on it: NewField, NewElement, Zero, Add, Mul (hash/emulated/bn254/poseidon/
# poseidon.go:28-165), Sub + IsZero (:123-126), AssertIsEqual (:224), Select-style muxes
# (tree/smt/emulated/utils.go).  The construction is gnark's:
#   * an element is 4 limbs of 64 bits with an overflow count: Add / Sub work limb by limb and only
#     grow the overflow (Sub adds a multiple of p whose limbs dominate the subtrahend's);
#   * Mul asks a hint for quotient k, remainder r and carries c with a b = k p + r, range-checks
#     their limbs (std/rangecheck) and queues the check; after Define ONE commitment to every limb
#     involved gives a challenge X (std/multicommit) and each queued product is verified as the
#     polynomial identity  a(X) b(X) = r(X) + k(X) p(X) + (2^64 - X) c(X)  over the native field;
#   * Reduce = Mul by one; AssertIsEqual = the same identity with r = 0 on a - b;
#   * an operand is reduced first when the product's columns could leave the native field.
# Limb products must stay below r / 8: overflow(a) + overflow(b) <= MAX_MUL_OVERFLOW.
NATIVE_BITS = 254
MAX_OVERFLOW = NATIVE_BITS - 2 - 64                   # additions: limb < 2^(64 + overflow)
MAX_MUL_OVERFLOW = NATIVE_BITS - 2 - 2 * 64 - 3 - 2   # = 119


class _El(Element):
    """Element with the bookkeeping of gnark's emulated.Element: overflow, internal (limb widths
    already enforced), constant value."""
    def __init__(self, limbs, params, overflow=0, internal=False, const=None):
        super().__init__(limbs, params)
        self.overflow, self.internal, self.const = overflow, internal, const


class Field:
    def __init__(self, api, params):
        from . import rangecheck
        if params.nb_limbs != 4 or params.bits_per_limb != 64 or \
                not 1 << 224 <= params.modulus < 1 << 256:
            raise ValueError("emulated.Field: 4 limbs of 64 bits, 2^224 <= modulus < 2^256")
        self.api, self.params, self.p = api, params, params.modulus
        self.rc = rangecheck.New(api)
        self.checks = []             # queued products: (a, b, r | None, k, c)
        self._enforced = set()
        self._zero = self._const_el(0)
        self._one = self._const_el(1)
        self._pads = {}
        api.Defer(self._perform_mul_checks)

    # ------------------------------------------------------------------ elements
    def _const_el(self, v):
        v %= self.p
        return _El(limbs_of(v, self.params), self.params, 0, True, v)

    def Zero(self):
        return self._zero

    def One(self):
        return self._one

    def Modulus(self):
        """emulated.Field.Modulus: p itself as an (unreduced) constant element"""
        m = (1 << 64) - 1
        return _El([(self.p >> (64 * i)) & m for i in range(4)], self.params, 0, True, 0)

    def NewElement(self, v):
        """int -> constant; Element (a witness' limbs) -> tracked element whose limb widths are
        enforced at first use."""
        if isinstance(v, _El):
            return v
        if isinstance(v, Element):
            return _El(v.Limbs, self.params, 0, False, None)
        return self._const_el(int(v))

    def _el(self, x):
        return x if isinstance(x, _El) else self.NewElement(x)

    def _enforce_width(self, e):
        """emulated.Field.enforceWidthConditional: limbs of a witness element are 64 bits wide, the
        top one as wide as the modulus' top limb."""
        if e.internal or id(e) in self._enforced:
            return
        self._enforced.add(id(e))
        e.internal = True
        top = self.p.bit_length() - 192
        for i, l in enumerate(e.Limbs):
            self.rc.Check(l, 64 if i < 3 else top)

    # ------------------------------------------------------------------ linear operations
    def Add(self, a, b):
        a, b = self._el(a), self._el(b)
        if a.const is not None and b.const is not None:
            return self._const_el(a.const + b.const)
        self._enforce_width(a)
        self._enforce_width(b)
        while max(a.overflow, b.overflow) + 1 > MAX_OVERFLOW:
            if a.overflow >= b.overflow:
                a = self.Reduce(a)
            else:
                b = self.Reduce(b)
        api = self.api
        return _El([api.Add(x, y) for x, y in zip(a.Limbs, b.Limbs)], self.params,
                   max(a.overflow, b.overflow) + 1, True)

    def _sub_padding(self, overflow):
        """a multiple of p whose limbs are all >= 2^(64 + overflow + 1) (gnark subPadding)."""
        pad = self._pads.get(overflow)
        if pad is None:
            top = 1 << (64 + overflow + 1)
            big = sum(top << (64 * i) for i in range(4))
            delta = (-big) % self.p
            m = (1 << 64) - 1
            pad = self._pads[overflow] = [top + ((delta >> (64 * i)) & m) for i in range(4)]
        return pad

    def Sub(self, a, b):
        a, b = self._el(a), self._el(b)
        if a.const is not None and b.const is not None:
            return self._const_el(a.const - b.const)
        self._enforce_width(a)
        self._enforce_width(b)
        while max(a.overflow, b.overflow + 2) + 1 > MAX_OVERFLOW:
            if a.overflow >= b.overflow + 2:
                a = self.Reduce(a)
            else:
                b = self.Reduce(b)
        api, pad = self.api, self._sub_padding(b.overflow)
        return _El([api.Sub(api.Add(x, q), y) for x, q, y in zip(a.Limbs, pad, b.Limbs)],
                   self.params, max(a.overflow, b.overflow + 2) + 1, True)

    def ModAdd(self, a, b, modulus):
        """emulated.Field.ModAdd(a, b, modulus): (a + b) mod modulus for a modulus given as an
        element.  The reference passes the field's own modulus (hash/emulated/bn254/mimc7/mimc.go:59):
        the sum, reduced; any other modulus is not restated."""
        m = self._el(modulus)
        if m.const is None or sum(int(l) << (64 * i) for i, l in enumerate(m.Limbs)) != self.p:
            raise NotImplementedError("ModAdd: only the field's own modulus")
        return self.Reduce(self.Add(a, b))

    def Neg(self, a):
        return self.Sub(self._zero, a)

    def Select(self, sel, a, b):
        a, b = self._el(a), self._el(b)
        self._enforce_width(a)
        self._enforce_width(b)
        api = self.api
        return _El([api.Select(sel, x, y) for x, y in zip(a.Limbs, b.Limbs)], self.params,
                   max(a.overflow, b.overflow), True)

    def Lookup2(self, b0, b1, a, b, c, d):
        """emulated.Field.Lookup2: a, b, c, d for (b1 b0) = 00, 01, 10, 11, limb by limb"""
        els = [self._el(x) for x in (a, b, c, d)]
        for e in els:
            self._enforce_width(e)
        api = self.api
        return _El([api.Lookup2(b0, b1, *ls) for ls in zip(*(e.Limbs for e in els))], self.params,
                   max(e.overflow for e in els), True)

    def Mux(self, sel, *inputs):
        """emulated.Field.Mux: inputs[sel], sel < len(inputs) (the reference passes 4 and 8 inputs,
        tree/smt/emulated/utils.go:26,33): selector bits, then a tree of Selects per limb."""
        els = [self._el(x) for x in inputs]
        n = len(els)
        if n == 1:
            return els[0]
        nbits = (n - 1).bit_length()
        api = self.api
        bits = api.ToBinary(sel, nbits)
        if n != 1 << nbits:
            api.AssertIsLessOrEqual(sel, n - 1)
        level = els
        for b in bits:
            nxt = []
            for i in range(0, len(level), 2):
                nxt.append(self.Select(b, level[i + 1], level[i]) if i + 1 < len(level) else level[i])
            level = nxt
        return level[0]

    # ------------------------------------------------------------------ products
    @staticmethod
    def _limbs_for_hint(e):
        """limbs and their maxima; a constant's zero top limbs are dropped"""
        if e.const is not None:
            ls = [int(l) for l in e.Limbs]
            while len(ls) > 1 and ls[-1] == 0:
                ls.pop()
            return ls, ls
        return list(e.Limbs), [(1 << (64 + e.overflow)) - 1] * len(e.Limbs)

    def _mul_mod(self, a, b, zero_remainder=False):
        api, p = self.api, self.p
        (al, am), (bl, bm) = self._limbs_for_hint(a), self._limbs_for_hint(b)
        amax = sum(v << (64 * i) for i, v in enumerate(am))
        bmax = sum(v << (64 * i) for i, v in enumerate(bm))
        kbits = max((amax * bmax // p).bit_length(), 1)
        nk = -(-kbits // 64)
        k, r, c = api.NewHintEmulMul(al, bl, p, nk)
        for i, l in enumerate(k):
            self.rc.Check(l, 64 if i < nk - 1 else kbits - 64 * (nk - 1))
        if zero_remainder:
            for l in r:
                api.AssertIsEqual(l, 0)
        else:
            top = p.bit_length() - 192
            for i, l in enumerate(r):
                self.rc.Check(l, 64 if i < 3 else top)
        # carries: |c_j| < 2^cb with the column maxima of both sides
        pl = [(p >> (64 * i)) & (2**64 - 1) for i in range(4)]
        m64 = (1 << 64) - 1
        km = [m64] * (nk - 1) + [(1 << (kbits - 64 * (nk - 1))) - 1]
        ncols = max(len(am) + len(bm) - 1, nk + 3)
        cmax = 0
        for j in range(ncols - 1):
            t = sum(am[i] * bm[j - i] for i in range(len(am)) if 0 <= j - i < len(bm))
            s = (m64 if j < 4 else 0) + sum(km[i] * pl[j - i] for i in range(nk) if 0 <= j - i < 4)
            cmax = (max(t, s) + cmax) >> 64
            cb = max(cmax.bit_length(), 1) + 1
            self.rc.Check(api.Add(c[j], 1 << cb), cb + 1)
        res = _El(r, self.params, 0, True)
        self.checks.append((a, b, None if zero_remainder else res, k, c))
        return res

    def _fit_for_mul(self, a, b):
        while a.overflow + b.overflow > MAX_MUL_OVERFLOW:
            if a.overflow >= b.overflow:
                a = self.Reduce(a)
            else:
                b = self.Reduce(b)
        return a, b

    def Mul(self, a, b):
        a, b = self._el(a), self._el(b)
        if a.const is not None and b.const is not None:
            return self._const_el(a.const * b.const)
        self._enforce_width(a)
        self._enforce_width(b)
        a, b = self._fit_for_mul(a, b)
        return self._mul_mod(a, b)

    MulMod = Mul

    def Reduce(self, a):
        a = self._el(a)
        self._enforce_width(a)
        if a.const is not None or a.overflow == 0:
            return a
        return self._mul_mod(a, self._one)

    def AssertIsEqual(self, a, b):
        a, b = self._el(a), self._el(b)
        if a.const is not None and b.const is not None:
            if a.const != b.const:
                raise ValueError("emulated constants differ")
            return
        d = self.Sub(b, a)
        if d.overflow > MAX_MUL_OVERFLOW:
            d = self.Reduce(d)
        self._mul_mod(d, self._one, zero_remainder=True)

    # ------------------------------------------------------------------ bits and flags
    def AssertIsInRange(self, a):
        """value < p, limb widths enforced: the bits of the limbs compared with p - 1."""
        a = self._el(a)
        if a.overflow:
            raise ValueError("AssertIsInRange: reduce first")
        self._enforce_width(a)
        api = self.api
        bits = []
        for i, l in enumerate(a.Limbs):
            bits += api.ToBinary(l, 64 if i < 3 else self.p.bit_length() - 192)
        api._must_be_less_or_eq_cst(bits, self.p - 1)
        return bits

    def ReduceStrict(self, a):
        r = self.Reduce(a)
        if r.const is None:
            self.AssertIsInRange(r)
        return r

    def ToBits(self, a):
        r = self.Reduce(a)
        if r.const is not None:
            return [(r.const >> i) & 1 for i in range(self.p.bit_length())]
        return self.AssertIsInRange(r)

    def IsZero(self, a):
        a = self._el(a)
        if a.const is not None:
            return 1 if a.const == 0 else 0
        r = self.ReduceStrict(a)
        api = self.api
        return api.IsZero(api.Add(api.Add(r.Limbs[0], r.Limbs[1]), api.Add(r.Limbs[2], r.Limbs[3])))

    # ------------------------------------------------------------------ deferred product checks
    def _perform_mul_checks(self, api):
        from . import multicommit
        if not self.checks:
            return
        checks, self.checks = self.checks, []
        rev = {}

        def wires_of(e):
            out = []
            if e is None or getattr(e, "const", None) is not None:
                return out
            for l in (e.Limbs if isinstance(e, Element) else e):
                v = api._v(l)
                out.extend(w for w in v.lc if w != 0)
            return out

        ws, seen = [], set()
        for a, b, r, k, c in checks:
            for e in (a, b, r, k, c):
                for w in wires_of(e):
                    if w not in seen:
                        seen.add(w)
                        ws.append(w)
        if not rev:
            rev.update({w: v for v, w in api.val_wire.items()})
        from ..frontend.api import Variable
        committed = [Variable({w: 1}, rev[w]) for w in ws]
        pl = [(self.p >> (64 * i)) & (2**64 - 1) for i in range(4)]

        def cb(api, x):
            powers = [1, x]
            cache = {}

            def power(i):
                while len(powers) <= i:
                    powers.append(api.Mul(powers[-1], x))
                return powers[i]

            def ev(limbs, const, key):
                """sum limbs_i x^i: free for constants (linear in the powers), Horner otherwise"""
                if key in cache:
                    return cache[key]
                limbs = list(limbs)
                if const:
                    acc = api.Sum([api.Mul(power(i), int(l)) for i, l in enumerate(limbs) if int(l)]) \
                        if any(int(l) for l in limbs) else 0
                else:
                    acc = limbs[-1]
                    for l in reversed(limbs[:-1]):
                        acc = api.Add(api.Mul(acc, x), l)
                cache[key] = acc
                return acc

            p_x = ev(pl, True, "p")
            shift = api.Sub(1 << 64, x)
            for a, b, r, k, c in checks:
                ea = ev(a.Limbs, a.const is not None, id(a))
                eb = ev(b.Limbs, b.const is not None, id(b))
                rhs = api.Add(api.Mul(ev(k, False, id(k)), p_x), api.Mul(ev(c, False, id(c)), shift))
                if r is not None:
                    rhs = api.Add(rhs, ev(r.Limbs, False, id(r)))
                api.AssertIsEqual(api.Mul(ea, eb), rhs)

        multicommit.WithCommitment(api, cb, *committed)


def NewField(api, params):
    """emulated.NewField[T](api): one Field per (builder, parameter set)."""
    fields = api.__dict__.setdefault("_emulated_fields", {})
    f = fields.get(params.name)
    if f is None:
        f = fields[params.name] = Field(api, params)
    return f


def ValueOf(x: int, params=Secp256k1Fp):
    """emulated.ValueOf[T]: assignment-side limbs."""
    return limbs_of(x, params)
