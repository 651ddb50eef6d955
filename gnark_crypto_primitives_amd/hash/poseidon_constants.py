"""Poseidon-BN254 (x^5, RF=8) parameter generation.

Generates, from first principles, the public parameter tables that the reference keeps as
decimal strings in hash/native/bn254/poseidon/constants.go:48 (strC), :2291 (strM),
:4414 (strS), :22733 (strP):

  * Grain-LFSR round constants and Cauchy MDS matrix exactly as published with the Poseidon
    paper (eprint 2019/458, reference script `generate_parameters_grain.sage`);
  * the "optimised" form used by circomlib/iden3 (`poseidon_constants_opt`): compressed round
    constants C, the pre-sparse matrix P and the per-partial-round sparse matrices S.

tests/test_poseidon.py asserts equality with the reference tables (as text, only when
/root/reference is present) for every width t = 2..17, and the circomlib known-answer vectors.
"""
import functools

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
N_ROUNDS_F = 8
# reference poseidon.go:119
N_ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68]


class Grain:
    def __init__(self, field, sbox, n, t, rf, rp):
        bits = []
        for val, w in ((field, 2), (sbox, 4), (n, 12), (t, 12), (rf, 10), (rp, 10)):
            bits += [(val >> (w - 1 - i)) & 1 for i in range(w)]
        bits += [1] * 30
        assert len(bits) == 80
        self.s = bits
        for _ in range(160):
            self._step()

    def _step(self):
        s = self.s
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb

    def bit(self):
        # self-shrinking: take pairs, emit second bit when first is 1
        while True:
            a = self._step()
            b = self._step()
            if a:
                return b

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def inv(x):
    return pow(x % R, R - 2, R)


def mat_mul(a, b):
    n, m, k = len(a), len(b[0]), len(b)
    return [[sum(a[i][l] * b[l][j] for l in range(k)) % R for j in range(m)] for i in range(n)]


def mat_vec_left(v, m):  # row-vector * matrix
    return [sum(v[i] * m[i][j] for i in range(len(v))) % R for j in range(len(m[0]))]


def mat_T(m):
    return [list(r) for r in zip(*m)]


def mat_inv(m):
    n = len(m)
    a = [list(row) + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(m)]
    for c in range(n):
        p = next(r for r in range(c, n) if a[r][c] % R)
        a[c], a[p] = a[p], a[c]
        iv = inv(a[c][c])
        a[c] = [x * iv % R for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % R for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


@functools.lru_cache(maxsize=None)
def raw_params(t):
    """Un-optimised Poseidon parameters: round constants [(RF+RP)*t] and MDS matrix M[t][t]."""
    rp = N_ROUNDS_P[t - 2]
    g = Grain(1, 0, 254, t, N_ROUNDS_F, rp)
    consts = []
    for _ in range((N_ROUNDS_F + rp) * t):
        v = g.bits(254)
        while v >= R:
            v = g.bits(254)
        consts.append(v)
    while True:
        rl = [g.bits(254) % R for _ in range(2 * t)]
        if len(set(rl)) != 2 * t:
            continue
        xs, ys = rl[:t], rl[t:]
        if any((x + y) % R == 0 for x in xs for y in ys):
            continue
        m = [[inv(xs[i] + ys[j]) for j in range(t)] for i in range(t)]
        return consts, m


@functools.lru_cache(maxsize=None)
def opt_params(t):
    """Optimised tables in the reference's layout (constants.go): returns (C, M, P, S) with
    C flat [8t+RP], M and P as t x t *in the reference's transposed orientation*
    (poseidon.go:213-224 computes out[i] = sum_j m[j][i]*in[j]), S flat [(2t-1)*RP]
    (poseidon.go:152-166)."""
    rp = N_ROUNDS_P[t - 2]
    rf2 = N_ROUNDS_F // 2
    consts, m = raw_params(t)
    c = [consts[i * t:(i + 1) * t] for i in range(N_ROUNDS_F + rp)]
    minv = mat_inv(m)

    def mv(mat, v):
        return [sum(mat[i][j] * v[j] for j in range(t)) % R for i in range(t)]

    # pull every round constant in front of the preceding linear layer; in the partial
    # section only coordinate 0 stays with its round, the rest folds into the round before
    partial = [0] * rp
    for i in range(rf2 + rp - 1, rf2 - 1, -1):      # i+1 = rf2+rp .. rf2+1
        k = mv(minv, c[i + 1])
        partial[i - rf2] = k[0]
        c[i] = [c[i][0]] + [(c[i][j] + k[j]) % R for j in range(1, t)]
    C = list(c[0])
    for i in range(1, rf2 + 1):
        C += mv(minv, c[i])
    C += partial
    for i in range(rf2 + rp + 1, N_ROUNDS_F + rp):
        C += mv(minv, c[i])
    assert len(C) == 8 * t + rp

    # sparse factorisation, last partial round first
    S = [None] * rp
    mmul = [row[:] for row in m]
    for r in range(rp - 1, -1, -1):
        mhat = [row[1:] for row in mmul[1:]]
        a = mmul[0][0]
        w = [mmul[i][0] for i in range(1, t)]
        v = mat_vec_left(mmul[0][1:], mat_inv(mhat))
        S[r] = [a] + v + w
        d = [[1] + [0] * (t - 1)] + [[0] + mhat[i] for i in range(t - 1)]
        mmul = mat_mul(d, m)
    Sflat = [x for row in S for x in row]
    return C, mat_T(m), mat_T(mmul), Sflat
