"""pyref -- pure-Python big-integer ORACLE (TEST INFRASTRUCTURE, not product code).

Independent restatement, with Python ints, of everything the hot path computes, used to pin the
C oracle (oracle/c) and the HIP path at sizes Python finishes in seconds:

  * BN254 fields and groups (G1: y^2 = x^3 + 3 over Fp; G2: y^2 = x^3 + 3/(9+u) over Fp2);
  * the optimal-ate pairing (tower Fp12 = Fp[w]/(w^12 - 18 w^6 + 82)), used to VERIFY proofs
    (gnark groth16.Verify, backend/groth16/bn254/verify.go [UPSTREAM-RECALL]);
  * textbook (un-optimised) Poseidon permutation from the Grain-LFSR parameters, pinned by the
    circomlib/iden3 known-answer vectors -- an algorithmically different
    path from the optimised gadget in hash/native/bn254/poseidon/poseidon.go:116-183;
  * BabyJubJub in gnark's reduced twisted-Edwards form (a = -1), pinned by K3;
  * Groth16: R1CS evaluation at the trapdoor and the closed-form expected proof;
  * Keccak-256 (legacy padding) on a flat 25-lane state with FIPS 202's published tables, pinned
    by hashlib.sha3_256 through the shared permutation, and secp256k1 -> Ethereum address, pinned
    by the public vectors for private keys 1 and 2 (config 5, ecdsa.DeriveAddress).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583

G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def inv(x, m):
    return pow(x % m, m - 2, m)


# ------------------------------------------------------------------ Fp2 -----------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = inv(a[0] * a[0] + a[1] * a[1], P)
    return (a[0] * d % P, -a[1] * d % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


F2_ZERO, F2_ONE = (0, 0), (1, 0)
B2 = f2_mul((3, 0), f2_inv((9, 1)))   # twist coefficient 3/(9+u)


# ------------------------------------------------------------------ groups (affine, None = inf)
class _G1:
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    inv = staticmethod(lambda a: inv(a, P))
    neg = staticmethod(lambda a: -a % P)


class _G2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, inv, neg = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_inv, f2_neg))


def _pt_add(F, p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if y1 != y2 or y1 == F.zero:
            return None
        three_x2 = F.mul(F.add(F.add(x1, x1), x1), x1)
        lam = F.mul(three_x2, F.inv(F.add(y1, y1)))
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)


def _pt_mul(F, p, k):
    acc = None
    while k:
        if k & 1:
            acc = _pt_add(F, acc, p)
        p = _pt_add(F, p, p)
        k >>= 1
    return acc


def g1_add(p, q):
    return _pt_add(_G1, p, q)


def g1_mul(p, k):
    return _pt_mul(_G1, p, k % R)


def g1_neg(p):
    return None if p is None else (p[0], -p[1] % P)


def g2_add(p, q):
    return _pt_add(_G2, p, q)


def g2_mul(p, k):
    return _pt_mul(_G2, p, k % R)


def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 3) % P == 0


def g2_on_curve(p):
    if p is None:
        return True
    x, y = p
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == F2_ZERO


# ------------------------------------------------------------------ Fp12 and the pairing -------
# Fp12 = Fp[w]/(w^12 - 18 w^6 + 82); u = w^6 - 9.
def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for i in range(22, 11, -1):
        c = t[i]
        if c:
            t[i - 6] += 18 * c
            t[i - 12] -= 82 * c
    return [x % P for x in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    # extended Euclid on polynomials over Fp
    mod = [82, 0, 0, 0, 0, 0, -18 % P, 0, 0, 0, 0, 0, 1]
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], mod
    while _poly_deg(low):
        dl, dh = _poly_deg(low), _poly_deg(high)
        # r = high / low
        r = [0] * 13
        temp = list(high)
        il = inv(low[dl], P)
        for i in range(dh - dl, -1, -1):
            q = temp[dl + i] * il % P
            r[i] = q
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - q * low[c]) % P
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    c = inv(low[0], P)
    return [x * c % P for x in lm[:12]]


def _f12_from_fp(x):
    return [x % P] + [0] * 11


def _f12_from_fp2(a):
    # a0 + a1*u with u = w^6 - 9
    return [(a[0] - 9 * a[1]) % P, 0, 0, 0, 0, 0, a[1] % P, 0, 0, 0, 0, 0]


_W2 = [0, 0, 1] + [0] * 9
_W3 = [0, 0, 0, 1] + [0] * 8


def _twist(q):
    return (f12_mul(_f12_from_fp2(q[0]), _W2), f12_mul(_f12_from_fp2(q[1]), _W3))


def _f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def _f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


class _G12:
    zero, one = [0] * 12, F12_ONE
    add, sub, mul, inv = map(staticmethod, (_f12_add, _f12_sub, f12_mul, f12_inv))
    neg = staticmethod(lambda a: [-x % P for x in a])


def _linefunc(p1, p2, t):
    (x1, y1), (x2, y2), (xt, yt) = p1, p2, t
    F = _G12
    if x1 != x2:
        m = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    elif y1 == y2:
        x1sq = F.mul(x1, x1)
        m = F.mul(F.add(F.add(x1sq, x1sq), x1sq), F.inv(F.add(y1, y1)))
    else:
        return F.sub(xt, x1)
    return F.sub(F.mul(m, F.sub(xt, x1)), F.sub(yt, y1))


ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE = 63


def miller_loop(q, p):
    """q in G2 (Fp2 affine), p in G1; returns the un-exponentiated Miller value in Fp12."""
    if q is None or p is None:
        return F12_ONE
    Q = _twist(q)
    Pt = (_f12_from_fp(p[0]), _f12_from_fp(p[1]))
    Rr = Q
    f = F12_ONE
    for i in range(LOG_ATE, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rr, Rr, Pt))
        Rr = _pt_add(_G12, Rr, Rr)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, _linefunc(Rr, Q, Pt))
            Rr = _pt_add(_G12, Rr, Q)
    Q1 = (f12_pow(Q[0], P), f12_pow(Q[1], P))
    nQ2 = (f12_pow(Q1[0], P), _G12.neg(f12_pow(Q1[1], P)))
    f = f12_mul(f, _linefunc(Rr, Q1, Pt))
    Rr = _pt_add(_G12, Rr, Q1)
    f = f12_mul(f, _linefunc(Rr, nQ2, Pt))
    return f


def final_exp(f):
    return f12_pow(f, (P ** 12 - 1) // R)


def pairing_product_is_one(pairs):
    """prod e(p_i, q_i) == 1 for pairs [(g1, g2), ...]"""
    f = F12_ONE
    for p, q in pairs:
        f = f12_mul(f, miller_loop(q, p))
    return final_exp(f) == F12_ONE


# ------------------------------------------------------------------ Poseidon (textbook form) ---
class _Grain:
    """Grain LFSR of the Poseidon paper (eprint 2019/458, generate_parameters_grain.sage)."""

    def __init__(self, t, rf, rp):
        bits = []
        for val, w in ((1, 2), (0, 4), (254, 12), (t, 12), (rf, 10), (rp, 10)):
            bits += [(val >> (w - 1 - i)) & 1 for i in range(w)]
        self.s = bits + [1] * 30
        for _ in range(160):
            self._next()

    def _next(self):
        s = self.s
        b = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(b)
        return b

    def field_bits(self):
        v = 0
        for _ in range(254):
            while not self._next():
                self._next()
            v = (v << 1) | self._next()
        return v


_RP = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]
_POSEIDON_CACHE = {}


def poseidon_params(t):
    if t not in _POSEIDON_CACHE:
        rp = _RP[t - 2]
        g = _Grain(t, 8, rp)
        rc = []
        while len(rc) < (8 + rp) * t:
            v = g.field_bits()
            if v < R:
                rc.append(v)
        xs = [g.field_bits() % R for _ in range(2 * t)]
        mds = [[inv(xs[i] + xs[t + j], R) for j in range(t)] for i in range(t)]
        _POSEIDON_CACHE[t] = (rp, rc, mds)
    return _POSEIDON_CACHE[t]


def poseidon_hash(inputs):
    """iden3/circomlib Poseidon(inputs): state = [0, inputs...], output state[0]."""
    t = len(inputs) + 1
    if not 2 <= t <= 17:
        raise ValueError("1..16 inputs")
    rp, rc, mds = poseidon_params(t)
    st = [0] + [x % R for x in inputs]
    for r in range(8 + rp):
        st = [(x + rc[r * t + i]) % R for i, x in enumerate(st)]
        if r < 4 or r >= 4 + rp:
            st = [pow(x, 5, R) for x in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(mds[i][j] * st[j] for j in range(t)) % R for i in range(t)]
    return st[0]


def poseidon_multihash(inputs):
    """davinci-node MultiPoseidon / reference poseidon.MultiHash (poseidon.go:54-91)."""
    if len(inputs) <= 16:
        return poseidon_hash(inputs)
    hashed = [poseidon_hash(inputs[i:i + 16]) for i in range(0, len(inputs), 16)]
    return poseidon_multihash(hashed)


# ------------------------------------------------------------------ BabyJubJub (reduced TE) ----
BJJ_A = R - 1
BJJ_D = 12181644023421730124874158521699555681764249180949974110617291017600649128846
BJJ_BASE = (9671717474070082183213120605117400219616337014328744928644933853176787189663,
            16950150798460657717958625567821834550301663161624707787222815936182638968203)
BJJ_ORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041


def bjj_on_curve(p):
    x, y = p
    return (BJJ_A * x * x + y * y - 1 - BJJ_D * x * x * y * y) % R == 0


def bjj_add(p, q):
    (x1, y1), (x2, y2) = p, q
    k = BJJ_D * x1 * x2 * y1 * y2 % R
    x3 = (x1 * y2 + y1 * x2) * inv(1 + k, R) % R
    y3 = (y1 * y2 - BJJ_A * x1 * x2) * inv(1 - k, R) % R
    return (x3, y3)


def bjj_mul(p, k):
    acc = (0, 1)
    while k:
        if k & 1:
            acc = bjj_add(acc, p)
        p = bjj_add(p, p)
        k >>= 1
    return acc


# ------------------------------------------------------------------ SMT (Arbo/circomlib) -------
def smt_root_from_path(key, value, siblings):
    """Fold a Merkle path bottom-up exactly as tree/smt/verifier_level.go:8-17 does when every
    level is 'top': leaf = H(key, value, 1); at level i (root = 0) bit i of key (LSB first,
    tree/smt/utils.go:11) selects (sibling, cur) vs (cur, sibling).  Levels below the insertion
    level (the deepest non-zero sibling, lev_ins.go:43-77) are skipped."""
    n = len(siblings)
    lev = max([i for i in range(n) if siblings[i]], default=-1) + 1   # insertion level
    cur = poseidon_hash([key, value, 1])
    for i in range(lev - 1, -1, -1):
        if (key >> i) & 1:
            cur = poseidon_hash([siblings[i], cur])
        else:
            cur = poseidon_hash([cur, siblings[i]])
    return cur


# ------------------------------------------------------------------ Keccak / secp256k1 address --
# FIPS 202 tables (round constants, rho offsets in pi order, pi lane order), flat index x + 5y.
_KECCAK_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_RHO = [1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61,
               20, 44]
_KECCAK_PI = [10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6,
              1]
_M64 = (1 << 64) - 1


def keccak_f1600(st):
    """In-place permutation of 25 lanes (the compact 'tiny' formulation: theta, then rho and pi as
    one walk along the pi cycle, chi row by row, iota)."""
    for rc in _KECCAK_RC:
        bc = [st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20] for i in range(5)]
        for i in range(5):
            t = bc[(i + 4) % 5] ^ (((bc[(i + 1) % 5] << 1) | (bc[(i + 1) % 5] >> 63)) & _M64)
            for j in range(0, 25, 5):
                st[j + i] ^= t
        t = st[1]
        for i in range(24):
            j = _KECCAK_PI[i]
            nxt = st[j]
            st[j] = ((t << _KECCAK_RHO[i]) | (t >> (64 - _KECCAK_RHO[i]))) & _M64
            t = nxt
        for j in range(0, 25, 5):
            row = st[j:j + 5]
            for i in range(5):
                st[j + i] = row[i] ^ (~row[(i + 1) % 5] & _M64 & row[(i + 2) % 5])
        st[0] ^= rc
    return st


def _sponge256(data: bytes, domain: int) -> bytes:
    rate = 136
    msg = bytearray(data) + bytes([domain]) + bytes(rate - 1 - len(data) % rate)
    msg[-1] |= 0x80
    st = [0] * 25
    for off in range(0, len(msg), rate):
        for k in range(rate // 8):
            st[k] ^= int.from_bytes(msg[off + 8 * k:off + 8 * k + 8], "little")
        keccak_f1600(st)
    return b"".join(x.to_bytes(8, "little") for x in st[:4])


def keccak256(data: bytes) -> bytes:
    """Legacy (pre-NIST, Ethereum) Keccak-256: domain byte 0x01."""
    return _sponge256(data, 0x01)


def sha3_256(data: bytes) -> bytes:
    """NIST SHA3-256 (domain byte 0x06): same sponge, comparable with hashlib."""
    return _sponge256(data, 0x06)


SECP_P = 2**256 - 2**32 - 977
SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
SECP_G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
          0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def secp256k1_mul(k, pt=SECP_G):
    """Jacobian double-and-add, MSB first (a = 0)."""
    X, Y, Z = 0, 1, 0
    for bit in bin(k % SECP_N)[2:]:
        if Z:
            S = 4 * X * Y * Y % SECP_P
            M = 3 * X * X % SECP_P
            X2 = (M * M - 2 * S) % SECP_P
            Y, Z = (M * (S - X2) - 8 * pow(Y, 4, SECP_P)) % SECP_P, 2 * Y * Z % SECP_P
            X = X2
        if bit == "1":
            if not Z:
                X, Y, Z = pt[0], pt[1], 1
                continue
            zz = Z * Z % SECP_P
            u2, s2 = pt[0] * zz % SECP_P, pt[1] * zz * Z % SECP_P
            h, r_ = (u2 - X) % SECP_P, (s2 - Y) % SECP_P
            assert h, "unsupported: doubling inside addition"
            hh = h * h % SECP_P
            hhh, v = h * hh % SECP_P, X * hh % SECP_P
            X3 = (r_ * r_ - hhh - 2 * v) % SECP_P
            Y, Z = (r_ * (v - X3) - Y * hhh) % SECP_P, Z * h % SECP_P
            X = X3
    zi = inv(Z, SECP_P)
    return X * zi * zi % SECP_P, Y * zi * zi * zi % SECP_P


def eth_address(pub) -> int:
    return int.from_bytes(keccak256(pub[0].to_bytes(32, "big") + pub[1].to_bytes(32, "big"))[12:],
                          "big")


# ------------------------------------------------------------------ Groth16 ---------------------
def root_of_unity(log_n):
    return pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)


def lagrange_at(tau, log_n, count):
    """L_k(tau) for k < count on the size-2^log_n domain."""
    n = 1 << log_n
    w = root_of_unity(log_n)
    zt = (pow(tau, n, R) - 1) % R
    out = []
    wk = 1
    ninv = inv(n, R)
    for _ in range(count):
        out.append(zt * wk % R * ninv % R * inv(tau - wk, R) % R)
        wk = wk * w % R
    return out


def qap_at(constraints, n_wires, tau, log_n):
    """(A_i(tau), B_i(tau), C_i(tau)) for every wire from the R1CS rows [(L, R, O, ...)]."""
    lag = lagrange_at(tau, log_n, len(constraints))
    A, B, C = [0] * n_wires, [0] * n_wires, [0] * n_wires
    for k, con in enumerate(constraints):
        lk = lag[k]
        for vec, lc in ((A, con[0]), (B, con[1]), (C, con[2])):
            for w, c in lc.items():
                vec[w] = (vec[w] + c * lk) % R
    return A, B, C


def expected_proof(constraints, n_wires, n_public, wires, trapdoor, log_n, r, s):
    """Closed form of the Groth16 proof from the trapdoor -- no NTT, no MSM:
    Ar = (alpha + A(tau) + r delta) G1, Bs = (beta + B(tau) + s delta) G2,
    Krs = (sum_priv w_i (beta A_i + alpha B_i + C_i)/delta + H(tau) Z(tau)/delta
           + s a + r b - r s delta) G1   with H = (A B - C)/Z."""
    tau, alpha, beta, gamma, delta = trapdoor
    A, B, C = qap_at(constraints, n_wires, tau, log_n)
    at = sum(w * x for w, x in zip(wires, A)) % R
    bt = sum(w * x for w, x in zip(wires, B)) % R
    ct = sum(w * x for w, x in zip(wires, C)) % R
    n = 1 << log_n
    zt = (pow(tau, n, R) - 1) % R
    ht = (at * bt - ct) * inv(zt, R) % R
    a = (alpha + at + r * delta) % R
    b = (beta + bt + s * delta) % R
    dinv = inv(delta, R)
    kpriv = sum(wires[i] * ((beta * A[i] + alpha * B[i] + C[i]) % R)
                for i in range(n_public, n_wires)) % R
    k = (kpriv * dinv + ht * zt % R * dinv + s * a + r * b - r * s % R * delta) % R
    return g1_mul(G1_GEN, a), g1_mul(G1_GEN, k), g2_mul(G2_GEN, b)


def verify(vk, public_wires, proof):
    """vk = dict(alpha=G1, beta=G2, gamma=G2, delta=G2, k=[G1...]); public_wires includes ONE.
    e(Ar, Bs) == e(alpha, beta) e(sum w_i K_i, gamma) e(Krs, delta)."""
    ar, krs, bs = proof
    if not (g1_on_curve(ar) and g1_on_curve(krs) and g2_on_curve(bs)):
        return False
    acc = None
    for w, kpt in zip(public_wires, vk["k"]):
        acc = g1_add(acc, g1_mul(kpt, w))
    return pairing_product_is_one([(ar, bs), (g1_neg(vk["alpha"]), vk["beta"]),
                                   (g1_neg(acc), vk["gamma"]), (g1_neg(krs), vk["delta"])])
