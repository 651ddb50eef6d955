"""Groth16 verifier over BN254 -- host side, like gnark's own (``groth16.Verify`` in
backend/groth16/bn254/verify.go runs on the CPU next to the accelerated prover too)
[UPSTREAM-RECALL].  It closes the loop for ``test.Assert.ProverSucceeded``: setup -> GPU prove ->
verify, without the test oracle.

Verification is not on the north-star hot path (one multi-pairing per proof, 256-byte inputs), so
this is plain Python integers: Fq12 as the single extension Fq[w] / (w^12 - 18 w^6 + 82), G2 points
carried into E(Fq12) through the sextic twist (x w^2, y w^3), optimal ate Miller loop with generic
chord-and-tangent lines, final exponentiation split into the easy part (p^6 - 1)(p^2 + 1) and the
hard part (p^4 - p^2 + 1) / r.  The check is gnark's:
    e(Ar, Bs) == e(alpha, beta) . e(sum_i pub_i K_i, gamma) . e(Krs, delta)
"""
from __future__ import annotations

import numpy as np

from .frontend.compile import array_to_ints

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP = 29793968203157093288                      # 6 t + 2
_MONT_INV = pow(1 << 256, P - 2, P)


# ---- Fq12 = Fq[w] / (w^12 - 18 w^6 + 82): coefficient lists of length 12 -------------------------
def _f12(c0=0):
    return [c0 % P] + [0] * 11


_ONE, _ZERO = _f12(1), _f12(0)


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):          # w^k = 18 w^(k-6) - 82 w^(k-12)
        v = t[k]
        if v:
            t[k - 6] += 18 * v
            t[k - 12] -= 82 * v
    return [v % P for v in t[:12]]


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_scalar(a, k):
    return [x * k % P for x in a]


def f12_pow(a, e):
    out, base = _ONE, a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """extended Euclid on polynomials over Fq against the modulus w^12 - 18 w^6 + 82"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [82, 0, 0, 0, 0, 0, P - 18, 0, 0, 0, 0, 0, 1]
    while _poly_deg(low):
        dl, dh = _poly_deg(low), _poly_deg(high)
        # r = high / low (polynomial quotient)
        r = [0] * 13
        temp = list(high)
        inv_lead = pow(low[dl], P - 2, P)
        for i in range(dh - dl, -1, -1):
            r[i] = temp[dl + i] * inv_lead % P
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - r[i] * low[c]) % P
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    k = pow(low[0], P - 2, P)
    return [x * k % P for x in lm[:12]]


def _fq2_to_f12(c0, c1):
    """a + b u with u^2 = -1, u = w^6 - 9"""
    out = _f12((c0 - 9 * c1) % P)
    out[6] = c1 % P
    return out


# ---- curve points over Fq12 (affine; None = infinity) ---------------------------------------------
def _double(pt):
    x, y = pt
    lam = f12_mul(f12_scalar(f12_mul(x, x), 3), f12_inv(f12_scalar(y, 2)))
    nx = f12_sub(f12_mul(lam, lam), f12_scalar(x, 2))
    ny = f12_sub(f12_mul(lam, f12_sub(x, nx)), y)
    return nx, ny


def _add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x1 == x2:
        return _double(p1) if y1 == y2 else None
    lam = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
    nx = f12_sub(f12_sub(f12_mul(lam, lam), x1), x2)
    ny = f12_sub(f12_mul(lam, f12_sub(x1, nx)), y1)
    return nx, ny


def _line(p1, p2, t):
    """the line through p1 and p2 (tangent if equal), evaluated at t"""
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def _twist(q):
    """E'(Fq2) -> E(Fq12): (x, y) -> (x w^2, y w^3)"""
    (x0, x1), (y0, y1) = q
    x, y = _fq2_to_f12(x0, x1), _fq2_to_f12(y0, y1)
    w2, w3 = [0] * 12, [0] * 12
    w2[2], w3[3] = 1, 1
    return f12_mul(x, w2), f12_mul(y, w3)


def miller_loop(q, p):
    """f_{6t+2, Q}(P) with the two Frobenius lines; q in E'(Fq2), p in E(Fq) (affine int pairs)"""
    if q is None or p is None:
        return _ONE
    Q = _twist(q)
    Pt = (_f12(p[0]), _f12(p[1]))
    Rp = Q
    f = _ONE
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f12_mul(f12_mul(f, f), _line(Rp, Rp, Pt))
        Rp = _double(Rp)
        if (ATE_LOOP >> i) & 1:
            f = f12_mul(f, _line(Rp, Q, Pt))
            Rp = _add(Rp, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), f12_scalar(f12_pow(Q1[1], P), P - 1))
    f = f12_mul(f, _line(Rp, Q1, Pt))
    Rp = _add(Rp, Q1)
    f = f12_mul(f, _line(Rp, nQ2, Pt))
    return f


def final_exponentiation(f):
    # easy part: f^((p^6 - 1)(p^2 + 1)); conjugation w -> -w is the p^6 Frobenius
    conj = [x if i % 2 == 0 else (P - x) % P for i, x in enumerate(f)]
    f = f12_mul(conj, f12_inv(f))
    f = f12_mul(f12_pow(f, P * P), f)
    return f12_pow(f, (P ** 4 - P * P + 1) // R)


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for [(g1 point, g2 point)]"""
    f = _ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(q, p))
    return final_exponentiation(f) == _ONE


# ---- G1 arithmetic for the public-input combination ------------------------------------------------
def _g1_add(a, b):
    if a is None or b is None:
        return a if b is None else b
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], P - 2, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], P - 2, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return x, (lam * (a[0] - x) - a[1]) % P


def _g1_mul(a, k):
    out = None
    k %= R
    while k:
        if k & 1:
            out = _g1_add(out, a)
        a = _g1_add(a, a)
        k >>= 1
    return out


def _g1_neg(a):
    return None if a is None else (a[0], (P - a[1]) % P)


def _on_g1(a):
    return a is None or (a[1] * a[1] - a[0] ** 3 - 3) % P == 0


_B2 = None


def _on_g2(q):
    global _B2
    if q is None:
        return True
    if _B2 is None:                                  # 3 / (9 + u)
        d = pow(82, P - 2, P)
        _B2 = (3 * 9 * d % P, (-3 * d) % P)
    (x0, x1), (y0, y1) = q
    mul2 = lambda a, b: ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    x3 = mul2(mul2((x0, x1), (x0, x1)), (x0, x1))
    y2 = mul2((y0, y1), (y0, y1))
    return ((y2[0] - x3[0] - _B2[0]) % P, (y2[1] - x3[1] - _B2[1]) % P) == (0, 0)


# ---- gnark memory image -> integers ----------------------------------------------------------------
def _unmont(words):
    return [v * _MONT_INV % P for v in array_to_ints(np.ascontiguousarray(words).reshape(-1, 4))]


def g1_from_image(a):
    v = _unmont(a)
    return None if not any(v) else (v[0], v[1])


def g2_from_image(a):
    v = _unmont(a)
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def _g2_add(a, b):
    """affine addition on the twist (Fq2 coordinates)"""
    if a is None or b is None:
        return a if b is None else b
    mul2 = lambda x, y: ((x[0] * y[0] - x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)
    sub2 = lambda x, y: ((x[0] - y[0]) % P, (x[1] - y[1]) % P)

    def inv2(x):
        d = pow((x[0] * x[0] + x[1] * x[1]) % P, P - 2, P)
        return (x[0] * d % P, (-x[1]) * d % P)
    if a[0] == b[0]:
        if ((a[1][0] + b[1][0]) % P, (a[1][1] + b[1][1]) % P) == (0, 0):
            return None
        x2 = mul2(a[0], a[0])
        lam = mul2(((3 * x2[0]) % P, (3 * x2[1]) % P), inv2(((2 * a[1][0]) % P, (2 * a[1][1]) % P)))
    else:
        lam = mul2(sub2(b[1], a[1]), inv2(sub2(b[0], a[0])))
    x = sub2(sub2(mul2(lam, lam), a[0]), b[0])
    return x, sub2(mul2(lam, sub2(a[0], x)), a[1])


def _in_g2_subgroup(q):
    """[r] Q == O.  BN254's twist has a large cofactor: a point on the curve is not necessarily in
    the order-r subgroup the pairing is defined on (gnark's decoder rejects such points)."""
    if q is None:
        return True
    acc, base, k = None, q, R
    while k:
        if k & 1:
            acc = _g2_add(acc, base)
        base = _g2_add(base, base)
        k >>= 1
    return acc is None


def _canonical(words):
    """every Fq coordinate of a point image is a reduced Montgomery residue"""
    return all(v < P for v in array_to_ints(np.ascontiguousarray(words).reshape(-1, 4)))


def verify(vk, public_inputs, proof, commitments=None, pok=None) -> bool:
    """groth16.Verify.  vk: groth16.VerifyingKey (numpy arrays, gnark's image); public_inputs: the
    public wire values as integers (without the ONE wire); proof: uint64[32] = Ar | Krs | Bs.
    Keys with the commitment extension (vk.commitment_wires): ``commitments`` = uint64[n, 8]
    Pedersen commitments (proof.Commitments), ``pok`` = uint64[8] (proof.CommitmentPok); the
    commitment wires' values are recomputed by hashing, the folded proof of knowledge is checked
    with one more pairing product, and the commitments join the public-input sum
    (gnark backend/groth16/bn254/verify.go [UPSTREAM-RECALL])."""
    from . import hash_to_field as h2f
    proof = np.ascontiguousarray(proof, dtype=np.uint64).reshape(32)
    if not _canonical(proof):
        return False
    ar, krs, bs = g1_from_image(proof[0:8]), g1_from_image(proof[8:16]), g2_from_image(proof[16:32])
    if not (_on_g1(ar) and _on_g1(krs) and _on_g2(bs)) or ar is None or bs is None:
        return False
    if not _in_g2_subgroup(bs):
        return False
    public_inputs = [int(x) for x in public_inputs]
    if any(not 0 <= x < R for x in public_inputs):
        return False                 # canonical encodings only: _g1_mul would reduce mod r
    ks = [g1_from_image(k) for k in vk.g1_k]
    n_com = len(getattr(vk, "commitment_wires", []))
    if len(public_inputs) + 1 + n_com != len(ks):
        raise ValueError(f"expected {len(ks) - 1 - n_com} public inputs, got {len(public_inputs)}")
    extra = []
    if n_com:
        if commitments is None or pok is None:
            return False
        commitments = np.ascontiguousarray(commitments, dtype=np.uint64).reshape(n_com, 8)
        pok = np.ascontiguousarray(pok, dtype=np.uint64).reshape(8)
        if not _canonical(commitments) or not _canonical(pok):
            return False
        ds = [g1_from_image(c) for c in commitments]
        pk_pt = g1_from_image(pok)
        if not all(_on_g1(d) for d in ds) or not _on_g1(pk_pt):
            return False
        # value of every commitment wire: hash of its commitment and of the public / earlier
        # commitment wires committed with it (wire w < nbPublic: public_inputs[w - 1])
        n_pub = len(public_inputs) + 1
        values = {}
        for i, d in enumerate(ds):
            hashed = [public_inputs[w - 1] if w < n_pub else values[w]
                      for w in vk.commitment_hashed[i]]
            values[vk.commitment_wires[i]] = h2f.commitment_challenge(d, hashed)
        extra = [values[w] for w in vk.commitment_wires]
        # folded proof of knowledge: e(sum_i c^i D_i, [-sigma] g) . e(pok, g) == 1
        ch = h2f.pok_challenge(extra)
        folded, cp = None, 1
        for d in ds:
            folded = _g1_add(folded, _g1_mul(d, cp) if cp != 1 else d)
            cp = cp * ch % R
        if not pairing_product_is_one([(folded, g2_from_image(vk.commitment_g_sigma_neg)),
                                       (pk_pt, g2_from_image(vk.commitment_g))]):
            return False
    vk_x = ks[0]
    for k, x in zip(ks[1:], public_inputs + extra):
        vk_x = _g1_add(vk_x, _g1_mul(k, int(x)))
    if n_com:
        for d in ds:
            vk_x = _g1_add(vk_x, d)
    pairs = [(_g1_neg(ar), bs),
             (g1_from_image(vk.g1_alpha), g2_from_image(vk.g2_beta)),
             (vk_x, g2_from_image(vk.g2_gamma)),
             (krs, g2_from_image(vk.g2_delta))]
    return pairing_product_is_one(pairs)
