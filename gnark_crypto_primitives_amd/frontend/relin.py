"""Critical-path rewrite of the linear part of a witness program.

The frontend records every Add / Sub / constant product as its own SSA operation, in the order the
gadget made the calls.  For the solver that order is a dependency chain: in a Poseidon partial
round the value entering the next S-box is  s0 * (x^5 + c) + s1 * st1 + s2 * st2, recorded as
ADDC, MULC, ADD, ADD after the S-box -- four more steps on a path that is already 3 products long,
160 levels x 65 rounds of them back to back.  None of these linear values is a wire (wires are
inputs, products and hint outputs); they are field elements the solver may compute in any way that
gives the same value.  This pass

* keeps every linear value symbolic (sum of coefficient * atom + constant; atoms = inputs,
  products, hint outputs, wire-backed values) until a non-linear operation, a constraint row or a
  wire needs it;
* materialises it there with the atoms ordered by how early they are available, so that only the
  LAST atom's term sits on the critical path, as one fused multiply-add:
      OP_FMAC  d = x * const + y        OP_FMA  d = x * y + z
* and, when that last atom is itself a product p * q that nothing else on the path waits for,
  computes  k * (p * q) + T  as  p * (k * q) + T  with k * q off the path: the S-box's last product
  and the multiplication by s0 become one step (the wire x^5 is still computed, beside the path).

Identical linear values are materialised once (the key is the normalised expression).  Field
arithmetic is exact, so every wire and every row keeps its value; tests/test_frontend.py runs the
rewritten, scheduled program against the sequential one.
"""
from __future__ import annotations

from .schedule import UNIT_HQ as sch_unit_hq, unit_outputs
from .api import (OP_ABC, OP_ADD, OP_ADDC, OP_BATCHINV, OP_BITS, OP_COMMIT, OP_COPY, OP_DIV, OP_HIST,
                  OP_HQ, OP_INV, OP_MUL, OP_MULABC, OP_MULC, OP_NEG, OP_PAIR, OP_SETC, OP_SUB, OP_XOR,
                  OP_XORABC, R)

OP_FMAC, OP_FMA = 18, 19          # (op, dst, x, const index, addend)  /  (op, dst, x, y, addend)
LINEAR = (OP_ADD, OP_SUB, OP_ADDC, OP_MULC, OP_NEG, OP_COPY, OP_SETC)
C_M, C_A = 10, 2                  # step costs, as in schedule.COST
MAX_TERMS = 24


def relinearize(ops, val_wire, consts, n_vals):
    """ops: post-DCE SSA ops (4-tuples).  Returns (new ops, n_vals'); FMA ops are 5-tuples.
    ``consts`` (list) is extended in place with the constants the rewrite needs."""
    cindex = {c: i for i, c in enumerate(consts)}

    def cid(c):
        c %= R
        i = cindex.get(c)
        if i is None:
            i = cindex[c] = len(consts)
            consts.append(c)
        return i

    out = []
    lin = {}                # symbolic linear values: val -> (terms {atom: coef}, const)
    depth = {}              # materialised values: availability (cost units); inputs: 0
    mul_of = {}             # product value -> (p, q)
    cache = {}
    nv = [n_vals]

    def fresh():
        nv[0] += 1
        return nv[0] - 1

    def dep(v):
        return depth.get(v, 0)

    def form(v):
        f = lin.get(v)
        return f if f is not None else ({v: 1}, 0)

    def emit(op, dst, a, b, z=None, cost=C_A):
        out.append((op, dst, a, b) if z is None else (op, dst, a, b, z))
        srcs = [a] if op in (OP_MULC, OP_ADDC, OP_NEG, OP_COPY, OP_FMAC) else \
            [] if op == OP_SETC else [a, b]
        if z is not None:
            srcs.append(z)
        depth[dst] = max([dep(s) for s in srcs], default=0) + cost
        return dst

    def materialize(v, target=None):
        """SSA value holding the linear value v (into `target` when it is wire-backed)"""
        terms, c0 = form(v)
        terms = {a: k % R for a, k in terms.items() if k % R}
        key = (tuple(sorted(terms.items())), c0 % R)
        if target is None:
            hit = cache.get(key)
            if hit is not None:
                return hit
            if len(terms) == 1 and c0 % R == 0 and next(iter(terms.values())) == 1:
                return next(iter(terms))
        items = sorted(terms.items(), key=lambda t: (dep(t[0]), t[0]))
        acc = None
        if c0 % R or not items:
            acc = emit(OP_SETC, fresh() if (items or target is None) else target, 0, cid(c0))
        for idx, (a, k) in enumerate(items):
            last = idx == len(items) - 1
            dst = target if (last and target is not None) else fresh()
            pq = mul_of.get(a)
            second = dep(items[idx - 1][0]) if idx else (dep(acc) if acc is not None else 0)
            if last and pq is not None and dep(a) >= second + C_M and (k != 1 or acc is not None):
                # k (p q) + T  ->  p' (k q') + T, p' = the later operand of the product
                p, q = pq if dep(pq[0]) >= dep(pq[1]) else (pq[1], pq[0])
                t = q if k == 1 else emit(OP_MULC, fresh(), q, cid(k), cost=C_M)
                acc = emit(OP_FMA, dst, p, t, acc, cost=C_M) if acc is not None else \
                    emit(OP_MUL, dst, p, t, cost=C_M)
            elif acc is None:
                acc = emit(OP_COPY, dst, a, 0) if k == 1 else emit(OP_MULC, dst, a, cid(k), cost=C_M)
            elif k == 1:
                acc = emit(OP_ADD, dst, a, acc)
            elif k == R - 1:
                acc = emit(OP_SUB, dst, acc, a)
            else:
                acc = emit(OP_FMAC, dst, a, cid(k), acc, cost=C_M)
        if target is not None and acc != target:       # constant-only wire value
            acc = emit(OP_COPY, target, acc, 0)
        if target is None:
            cache[key] = acc
        return acc

    def use(v):
        return materialize(v) if v in lin else v

    def combine(fa, ka, fb, kb, c=0):
        terms = {}
        for f, k in ((fa, ka), (fb, kb)):
            if f is None:
                continue
            for a, co in f[0].items():
                terms[a] = (terms.get(a, 0) + k * co) % R
            c += k * f[1]
        return ({a: k for a, k in terms.items() if k}, c % R)

    i, n = 0, len(ops)
    while i < n:
        op, dst, a, b = ops[i]
        i += 1
        if op in LINEAR:
            C = consts
            if op == OP_ADD:
                f = combine(form(a), 1, form(b), 1)
            elif op == OP_SUB:
                f = combine(form(a), 1, form(b), R - 1)
            elif op == OP_ADDC:
                f = combine(form(a), 1, None, 0, C[b])
            elif op == OP_MULC:
                f = combine(form(a), C[b], None, 0)
            elif op == OP_NEG:
                f = combine(form(a), R - 1, None, 0)
            elif op == OP_COPY:
                f = combine(form(a), 1, None, 0)
            else:
                f = ({}, C[b] % R)
            if len(f[0]) > MAX_TERMS:               # keep expressions bounded: split at the operands
                xa = use(a)
                xb = use(b) if op in (OP_ADD, OP_SUB) else None
                f = combine(({xa: 1}, 0), 1, ({xb: 1}, 0) if xb is not None else None,
                            (1 if op == OP_ADD else R - 1) if xb is not None else 0)
                if op == OP_ADDC:
                    f = (f[0], (f[1] + C[b]) % R)
                elif op == OP_MULC:
                    f = ({x: k * C[b] % R for x, k in f[0].items()}, f[1] * C[b] % R)
                elif op == OP_NEG:
                    f = ({x: (R - k) % R for x, k in f[0].items()}, (R - f[1]) % R)
            lin[dst] = f
            if dst in val_wire:                      # a wire: its slot must hold the value
                materialize(dst, target=dst)
                del lin[dst]
            continue
        if op == OP_BATCHINV:
            pairs = []
            for _ in range(dst):
                o = ops[i]
                i += 1
                pairs.append((o[0], o[1], use(o[2]), 0))   # sources may be linear values: materialise
            # materialisations emitted by use() come first, then the unit and its rows back to back
            out.append((op, dst, a, b))
            out.extend(pairs)
            d0 = max([dep(o[2]) for o in pairs], default=0) + 4000
            for o in pairs:
                depth[o[1]] = d0
            continue
        if op in sch_unit_hq:
            # operand rows: materialised values; the unit defines wire-backed values only
            rows = [use(ops[i + j][2]) for j in range(a)]
            i += a
            out.append((op, dst, a, b))
            out.extend((OP_HQ, 0, x, 0) for x in rows)
            d = max([dep(x) for x in rows], default=0) + \
                (4000 if op == OP_HIST else 20000 if op == OP_COMMIT else 150)
            for v in unit_outputs((op, dst, a, b)):
                depth[v] = d
            continue
        if op == OP_ABC:
            out.append((op, use(dst), use(a), use(b)))
            continue
        if op == OP_BITS:
            xa = use(a)
            out.append((op, dst, xa, b))
            for v in range(dst, dst + (b & 0xffff)):
                depth[v] = dep(xa) + 40
            continue
        if op in (OP_INV,):
            xa = use(a)
            out.append((op, dst, xa, b))
            depth[dst] = dep(xa) + 4000
            continue
        # two-operand non-linear ops
        xa, xb = use(a), use(b)
        out.append((op, dst, xa, xb))
        depth[dst] = max(dep(xa), dep(xb)) + (4000 if op == OP_DIV else C_M)
        if op in (OP_MUL, OP_MULABC):
            mul_of[dst] = (xa, xb)
    return out, nv[0]
