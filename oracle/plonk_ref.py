"""CPU restatement of the PLONK prover / verifier the GPU path implements -- TEST INFRASTRUCTURE
(only tests/ may import it).

gnark's ``plonk.Prove`` (backend/plonk/bn254/prove.go, third-party module go.mod:8) is the
interface this stands in for [UPSTREAM-RECALL]: KZG commitments over BN254, wire polynomials blinded
with multiples of Z_H, grand-product permutation argument, quotient split in three, linearisation
and two batched openings, Fiat-Shamir challenges from SHA-256.  The reference holds no PLONK vector
(its only test of the config-5 circuit proves with Groth16, ecc/secp256k1/ecdsa/address_test.go:57),
gnark cannot be built offline and its exact transcript / quotient layout are restated from memory:
**parity unpinned** -- this oracle pins the GPU path to an independent plain-integer implementation
of the SAME protocol, and the pairing check pins both to KZG soundness.

Plain Python integers and O(n log n) NTTs; commitments through the C oracle's MSM.
"""
from __future__ import annotations

import hashlib

import numpy as np

from oracle import cref, pyref

R = pyref.R
K = (1, 5, 25)            # column cosets of the identity permutation


def inv(x):
    return pow(x % R, R - 2, R)


def root(log_n):
    return pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)


def ntt(vals, log_n, inverse=False):
    """values <-> coefficients over the 2^log_n subgroup (natural order both sides)."""
    n = 1 << log_n
    a = list(vals) + [0] * (n - len(vals))
    w = root(log_n)
    if inverse:
        w = inv(w)
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, R)
        for s in range(0, n, length):
            x = 1
            for k in range(length // 2):
                u, v = a[s + k], a[s + k + length // 2] * x % R
                a[s + k], a[s + k + length // 2] = (u + v) % R, (u - v) % R
                x = x * wl % R
        length <<= 1
    if inverse:
        ni = inv(n)
        a = [x * ni % R for x in a]
    return a


def coset_eval(coeffs, log_m, g=5):
    """evaluations on g * <omega_m>, m = 2^log_m"""
    m = 1 << log_m
    c = list(coeffs) + [0] * (m - len(coeffs))
    x = 1
    for i in range(m):
        c[i] = c[i] * x % R
        x = x * g % R
    return ntt(c, log_m)


def coset_interp(evals, log_m, g=5):
    c = ntt(evals, log_m, inverse=True)
    gi, x = inv(g), 1
    for i in range(len(c)):
        c[i] = c[i] * x % R
        x = x * gi % R
    return c


def poly_eval(c, x):
    acc = 0
    for v in reversed(c):
        acc = (acc * x + v) % R
    return acc


def div_linear(c, z):
    """c(X) / (X - z), exact division (remainder dropped)"""
    q = [0] * (len(c) - 1)
    carry = 0
    for k in range(len(c) - 1, 0, -1):
        carry = (c[k] + z * carry) % R
        q[k - 1] = carry
    return q


# ---- transcript ---------------------------------------------------------------------------------------
def challenge(label: str, *parts) -> int:
    """SHA-256(label || parts): ints as 32 big-endian bytes, points as their uncompressed bytes
    (x || y big-endian, infinity = zeros), reduced mod r."""
    h = hashlib.sha256()
    h.update(label.encode())
    for p in parts:
        if isinstance(p, (bytes, bytearray)):
            h.update(p)
        elif p is None:
            h.update(bytes(64))
        elif isinstance(p, tuple):
            h.update(p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big"))
        else:
            h.update((int(p) % R).to_bytes(32, "big"))
    return int.from_bytes(h.digest(), "big") % R


# ---- setup ---------------------------------------------------------------------------------------------
def mont(xs):
    return np.frombuffer(b"".join((x % R * ((1 << 256) % R) % R).to_bytes(32, "little") for x in xs),
                         dtype=np.uint64).reshape(-1, 4).copy()


def g1_from_image(a):
    mi = pow(1 << 256, pyref.P - 2, pyref.P)
    v = [int.from_bytes(a.reshape(-1, 4)[i].tobytes(), "little") * mi % pyref.P for i in range(2)]
    return None if not any(v) else (v[0], v[1])


def commit(srs_img, coeffs):
    """KZG commitment sum_i c_i [tau^i]_1 through the C oracle's MSM"""
    n = len(coeffs)
    return g1_from_image(cref.msm(1, srs_img[:n], mont(coeffs)))


def setup(scs, seed):
    """SRS from a seeded tau (test only), selector / permutation polynomials and their commitments."""
    import random
    tau = random.Random(seed).randrange(2, R)
    n, log_n = 1 << scs.log_n, scs.log_n
    g1 = cref.fq_to_mont(np.frombuffer((1).to_bytes(32, "little") + (2).to_bytes(32, "little"),
                                       dtype=np.uint64).reshape(2, 4)).reshape(-1)
    pw, t = [], 1
    for _ in range(n + 6):
        pw.append(t)
        t = t * tau % R
    srs_img = cref.batch_mul(1, g1, mont(pw))
    w = root(log_n)
    ident = []
    for c in range(3):
        x = K[c]
        for _ in range(n):
            ident.append(x)
            x = x * w % R
    sig = [[ident[int(scs.sigma[c * n + r])] for r in range(n)] for c in range(3)]
    lag = {"ql": scs.qL, "qr": scs.qR, "qo": scs.qO, "qm": scs.qM, "qc": scs.qC,
           "s1": sig[0], "s2": sig[1], "s3": sig[2]}
    lag = {k: [int(x) % R for x in v] for k, v in lag.items()}
    coef = {k: ntt(v, log_n, inverse=True) for k, v in lag.items()}
    com = {k: commit(srs_img, v) for k, v in coef.items()}
    g2_tau = pyref.g2_mul(((10857046999023057135944570762232829481370756359578518086990519993285655852781,
                            11559732032986387107991004021392285783925812861821192530917403151452391805634),
                           (8495653923123431417604973247489272438418190587263600148770280649306958101930,
                            4082367875863433681332203403145435568316851327593401208105741076214120093531)),
                          tau)
    return {"log_n": log_n, "n_pub": scs.n_public - 1, "srs": srs_img, "lag": lag, "coef": coef,
            "com": com, "g2_tau": g2_tau, "tau": tau}


def vk_digest(key):
    parts = [key["log_n"], key["n_pub"]] + [key["com"][k] for k in
                                            ("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3")]
    return challenge("vk", *parts).to_bytes(32, "big")


# ---- prover ----------------------------------------------------------------------------------------------
def blinded_poly(vals, bs, log_n):
    """coefficients of the interpolant of `vals` on the domain + (bs[0] X^(k-1) + ... + bs[-1]) (X^n - 1)"""
    n = 1 << log_n
    cf = ntt(list(vals) + [0] * (n - len(vals)), log_n, inverse=True) + [0] * len(bs)
    for j, bj in enumerate(reversed(bs)):          # bj multiplies X^j
        cf[j] = (cf[j] - bj) % R
        cf[n + j] = (cf[n + j] + bj) % R
    return cf


def round1_commitments(srs_img, log_n, a, b, c, blind):
    """[a], [b], [c] of the prover's first round alone (for circuits too large to run the whole
    Python prover on): blinded wire polynomials committed through the C oracle's MSM."""
    return tuple(commit(srs_img, blinded_poly(col, blind[2 * i:2 * i + 2], log_n))
                 for i, col in enumerate((a, b, c)))


def prove(key, a, b, c, public, blind):
    """a, b, c: gate columns (ints); public: public inputs; blind: 9 scalars.  Returns the proof
    dict {a, b, c, z, tlo, tmid, thi, wz, wzw (points); ev = (a, b, c, s1, s2, zw) at zeta}."""
    log_n = key["log_n"]
    n = 1 << log_n
    w = root(log_n)
    srs, lag, coef = key["srs"], key["lag"], key["coef"]

    def blinded(vals, bs):
        return blinded_poly(vals, bs, log_n)
    ca, cb, cc = blinded(a, blind[0:2]), blinded(b, blind[2:4]), blinded(c, blind[4:6])
    A, B, C = commit(srs, ca), commit(srs, cb), commit(srs, cc)
    vkd = vk_digest(key)
    gamma = challenge("gamma", vkd, *public, A, B, C)
    beta = challenge("beta", gamma)
    # ---- permutation grand product
    col = [list(a) + [0] * (n - len(a)), list(b) + [0] * (n - len(b)), list(c) + [0] * (n - len(c))]
    sig = [lag["s1"], lag["s2"], lag["s3"]]
    z = [1] * n
    x = 1
    for i in range(n - 1):
        num = den = 1
        for k in range(3):
            num = num * (col[k][i] + beta * K[k] * x + gamma) % R
            den = den * (col[k][i] + beta * sig[k][i] + gamma) % R
        z[i + 1] = z[i] * num % R * inv(den) % R
        x = x * w % R
    cz = blinded(z, blind[6:9])
    Z = commit(srs, cz)
    alpha = challenge("alpha", beta, Z)
    # ---- quotient on the coset 5 * <omega_4n>
    lm = log_n + 2
    m = 1 << lm
    ea, eb, ec, ez = (coset_eval(p, lm) for p in (ca, cb, cc, cz))
    pi_l = [(-v) % R for v in public] + [0] * (n - len(public))
    epi = coset_eval(ntt(pi_l, log_n, inverse=True), lm)
    esel = {k: coset_eval(v, lm) for k, v in coef.items()}
    w4 = root(lm)
    tq = [0] * m
    ninv = inv(n)
    x = 5
    zh_inv = [inv(pow(5 * pow(w4, j, R) % R, n, R) - 1) for j in range(4)]
    for j in range(m):
        gate = (esel["ql"][j] * ea[j] + esel["qr"][j] * eb[j] + esel["qo"][j] * ec[j] +
                esel["qm"][j] * ea[j] % R * eb[j] + esel["qc"][j] + epi[j]) % R
        p1 = (ea[j] + beta * x + gamma) * (eb[j] + beta * 5 * x + gamma) % R * \
            (ec[j] + beta * 25 * x + gamma) % R * ez[j] % R
        p2 = (ea[j] + beta * esel["s1"][j] + gamma) * (eb[j] + beta * esel["s2"][j] + gamma) % R * \
            (ec[j] + beta * esel["s3"][j] + gamma) % R * ez[(j + 4) % m] % R
        zh = (pow(x, n, R) - 1) % R
        l1 = zh * ninv % R * inv(x - 1) % R
        num = (gate + alpha * (p1 - p2) + alpha * alpha % R * (ez[j] - 1) % R * l1) % R
        tq[j] = num * zh_inv[j % 4] % R
        x = x * w4 % R
    ct = coset_interp(tq, lm)
    assert not any(ct[3 * n + 6:]), "quotient degree too high: the witness does not satisfy the system"
    tlo, tmid, thi = ct[:n + 2], ct[n + 2:2 * n + 4], ct[2 * n + 4:3 * n + 6]
    TLO, TMID, THI = commit(srs, tlo), commit(srs, tmid), commit(srs, thi)
    zeta = challenge("zeta", alpha, TLO, TMID, THI)
    # ---- evaluations
    ev = (poly_eval(ca, zeta), poly_eval(cb, zeta), poly_eval(cc, zeta),
          poly_eval(coef["s1"], zeta), poly_eval(coef["s2"], zeta), poly_eval(cz, zeta * w % R))
    v = challenge("v", zeta, *ev)
    sc = lin_scalars(key, public, beta, gamma, alpha, zeta, ev)
    # ---- linearisation + openings
    L = n + 3
    pad = lambda p: list(p) + [0] * (L - len(p))
    r = [0] * L
    for name, s in (("qm", sc["qm"]), ("ql", sc["ql"]), ("qr", sc["qr"]), ("qo", sc["qo"]),
                    ("qc", 1), ("s3", sc["s3"])):
        for i, x_ in enumerate(coef[name]):
            r[i] = (r[i] + s * x_) % R
    for i, x_ in enumerate(cz):
        r[i] = (r[i] + sc["z"] * x_) % R
    for p, s in ((tlo, sc["tlo"]), (tmid, sc["tmid"]), (thi, sc["thi"])):
        for i, x_ in enumerate(p):
            r[i] = (r[i] + s * x_) % R
    r[0] = (r[0] + sc["r0"]) % R
    assert poly_eval(r, zeta) == 0
    num = list(r)
    vp = 1
    for p, e in ((ca, ev[0]), (cb, ev[1]), (cc, ev[2]), (coef["s1"], ev[3]), (coef["s2"], ev[4])):
        vp = vp * v % R
        pp = pad(p)
        for i in range(L):
            num[i] = (num[i] + vp * pp[i]) % R
        num[0] = (num[0] - vp * e) % R
    wz = div_linear(num, zeta)
    nz = list(cz)
    nz[0] = (nz[0] - ev[5]) % R
    wzw = div_linear(nz, zeta * w % R)
    return {"a": A, "b": B, "c": C, "z": Z, "tlo": TLO, "tmid": TMID, "thi": THI,
            "wz": commit(srs, wz), "wzw": commit(srs, wzw), "ev": ev}


def setup_fast(scs, srs_img, com, g2_tau):
    """Oracle key for ``prove_fast`` at sizes where ``setup``'s Python NTTs are too slow: selector
    and permutation polynomials from the circuit (Lagrange values built here, inverse NTTs in the C
    oracle); the SRS image, the eight key commitments and [tau]_2 are taken as data (they are compared
    with ``setup``'s on the small circuits of tests/test_gpu_plonk.py)."""
    import ctypes as C
    n, log_n = 1 << scs.log_n, scs.log_n
    w = root(log_n)
    ident = []
    for c in range(3):
        x = K[c]
        for _ in range(n):
            ident.append(x)
            x = x * w % R
    sig = [[ident[int(scs.sigma[c * n + r])] for r in range(n)] for c in range(3)]
    lag = {"ql": scs.qL, "qr": scs.qR, "qo": scs.qO, "qm": scs.qM, "qc": scs.qC,
           "s1": sig[0], "s2": sig[1], "s3": sig[2]}
    lag_m = {k: mont([int(x) % R for x in v]) for k, v in lag.items()}
    coef_m = {}
    for k, v in lag_m.items():
        cf = v.copy()
        cref.lib().zkref_ntt(cf.ctypes.data_as(C.c_void_p), log_n, 1, 0)
        coef_m[k] = cf
    return {"log_n": log_n, "n_pub": scs.n_public - 1, "srs": srs_img, "com": com, "g2_tau": g2_tau,
            "coef_m": coef_m, "sig_m": np.concatenate([lag_m[k] for k in ("s1", "s2", "s3")]),
            "sel_m": np.concatenate([coef_m[k] for k in
                                     ("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3")])}


# ---- the same prover with its per-element loops in C (oracle/c/zkref_plonk.inc) -----------------------
def _unmont(arr):
    return cref_ints(cref.fr_from_mont(arr))


def cref_ints(arr):
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int.from_bytes(a[i].tobytes(), "little") for i in range(a.shape[0])]


def _m1(x):
    return mont([x])[0]


def prove_fast(key, a, b, c, public, blind):
    """``prove`` with NTTs, grand product, quotient, evaluations and divisions in the C oracle; the
    protocol, the transcript and every formula are those of ``prove`` above (tests/test_plonk.py
    compares the two on small circuits), so BASELINE config 5 at 2^18 gates takes seconds."""
    import ctypes as C
    lib = cref.lib()
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    log_n = key["log_n"]
    n = 1 << log_n
    w = root(log_n)
    srs = key["srs"]
    if "coef_m" not in key:
        key["coef_m"] = {k: mont(v) for k, v in key["coef"].items()}
        key["sig_m"] = np.concatenate([mont(key["lag"][k]) for k in ("s1", "s2", "s3")])
        key["sel_m"] = np.concatenate([key["coef_m"][k] for k in
                                       ("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3")])
    coef_m = key["coef_m"]

    def blinded(vals_m, bs):
        cf = np.zeros((n + len(bs), 4), np.uint64)
        cf[:len(vals_m)] = vals_m
        head = np.ascontiguousarray(cf[:n])
        lib.zkref_ntt(P(head), log_n, 1, 0)
        cf[:n] = head
        for j, bj in enumerate(reversed(bs)):
            lo = _unmont(cf[j:j + 1])[0]
            cf[j] = _m1(lo - bj)
            cf[n + j] = _m1(bj)
        return cf

    def commit_m(cf):
        return g1_from_image(cref.msm(1, srs[:len(cf)], cf))

    def commit_many(cfs):
        """independent commitments side by side (ctypes releases the GIL around the C MSM)"""
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(cfs)) as ex:
            return list(ex.map(commit_m, cfs))

    def ev_at(cf, x):
        out = np.zeros(4, np.uint64)
        xm = _m1(x)
        lib.zkref_poly_eval(P(np.ascontiguousarray(cf)), C.c_size_t(len(cf)), P(xm), P(out))
        return _unmont(out)[0]
    am, bm, cm = (mont(list(v)) for v in (a, b, c))
    ca, cb, cc = blinded(am, blind[0:2]), blinded(bm, blind[2:4]), blinded(cm, blind[4:6])
    A, B, Cc = commit_many([ca, cb, cc])
    vkd = vk_digest(key)
    gamma = challenge("gamma", vkd, *public, A, B, Cc)
    beta = challenge("beta", gamma)
    cols = np.zeros((3, n, 4), np.uint64)
    for k, v in enumerate((am, bm, cm)):
        cols[k, :len(v)] = v
    z = np.zeros((n, 4), np.uint64)
    lib.zkref_plonk_z(P(cols), P(key["sig_m"]), P(_m1(beta)), P(_m1(gamma)), log_n, P(z))
    cz = blinded(z, blind[6:9])
    Z = commit_m(cz)
    alpha = challenge("alpha", beta, Z)
    pi_l = np.zeros((n, 4), np.uint64)
    if public:
        pi_l[:len(public)] = mont([(-v) % R for v in public])
    lib.zkref_ntt(P(pi_l), log_n, 1, 0)
    ct = np.zeros((4 * n, 4), np.uint64)
    lib.zkref_plonk_quotient(P(ca), P(cb), P(cc), C.c_size_t(len(ca)), P(cz), C.c_size_t(len(cz)),
                             P(pi_l), P(key["sel_m"]), P(_m1(beta)), P(_m1(gamma)), P(_m1(alpha)),
                             log_n, P(ct))
    assert not ct[3 * n + 6:].any(), "quotient degree too high: the witness does not satisfy the system"
    tlo, tmid, thi = (np.ascontiguousarray(x) for x in
                      (ct[:n + 2], ct[n + 2:2 * n + 4], ct[2 * n + 4:3 * n + 6]))
    TLO, TMID, THI = commit_many([tlo, tmid, thi])
    zeta = challenge("zeta", alpha, TLO, TMID, THI)
    ev = (ev_at(ca, zeta), ev_at(cb, zeta), ev_at(cc, zeta), ev_at(coef_m["s1"], zeta),
          ev_at(coef_m["s2"], zeta), ev_at(cz, zeta * w % R))
    v = challenge("v", zeta, *ev)
    sc = lin_scalars(key, public, beta, gamma, alpha, zeta, ev)
    L = n + 3
    r = np.zeros((L, 4), np.uint64)

    def axpy(dst, p, s):
        p = np.ascontiguousarray(p)
        lib.zkref_poly_axpy(P(dst), P(p), C.c_size_t(len(p)), P(_m1(s)))
    for name, s_ in (("qm", sc["qm"]), ("ql", sc["ql"]), ("qr", sc["qr"]), ("qo", sc["qo"]),
                     ("qc", 1), ("s3", sc["s3"])):
        axpy(r, coef_m[name], s_)
    axpy(r, cz, sc["z"])
    for p_, s_ in ((tlo, sc["tlo"]), (tmid, sc["tmid"]), (thi, sc["thi"])):
        axpy(r, p_, s_)
    r[0] = _m1(_unmont(r[0:1])[0] + sc["r0"])
    assert ev_at(r, zeta) == 0
    num = r.copy()
    vp, shift = 1, 0
    for p_, e in ((ca, ev[0]), (cb, ev[1]), (cc, ev[2]), (coef_m["s1"], ev[3]), (coef_m["s2"], ev[4])):
        vp = vp * v % R
        axpy(num, p_, vp)
        shift = (shift + vp * e) % R
    num[0] = _m1(_unmont(num[0:1])[0] - shift)
    wz = np.zeros((L - 1, 4), np.uint64)
    lib.zkref_div_linear(P(num), C.c_size_t(L), P(_m1(zeta)), P(wz))
    nz = cz.copy()
    nz[0] = _m1(_unmont(nz[0:1])[0] - ev[5])
    wzw = np.zeros((len(nz) - 1, 4), np.uint64)
    lib.zkref_div_linear(P(nz), C.c_size_t(len(nz)), P(_m1(zeta * w % R)), P(wzw))
    WZ, WZW = commit_many([wz, wzw])
    return {"a": A, "b": B, "c": Cc, "z": Z, "tlo": TLO, "tmid": TMID, "thi": THI,
            "wz": WZ, "wzw": WZW, "ev": ev}


def lin_scalars(key, public, beta, gamma, alpha, zeta, ev):
    """per-proof scalars of the linearisation polynomial
    r(X) = qm.qM + ql.qL + qr.qR + qo.qO + qC + s3.S3 + z.z(X) + tlo.t_lo + tmid.t_mid + thi.t_hi + r0"""
    n = 1 << key["log_n"]
    ea, eb, ec, es1, es2, ezw = ev
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * inv(n) % R * inv(zeta - 1) % R
    w = root(key["log_n"])
    pi = 0
    for j, x in enumerate(public):               # PI(zeta) = sum_j -x_j L_j(zeta)
        wj = pow(w, j, R)
        pi = (pi - x * (zh * wj % R * inv(n) % R * inv(zeta - wj) % R)) % R
    a1 = (ea + beta * zeta + gamma) * (eb + 5 * beta * zeta + gamma) % R * \
        (ec + 25 * beta * zeta + gamma) % R
    a2 = (ea + beta * es1 + gamma) * (eb + beta * es2 + gamma) % R
    zn2 = pow(zeta, n + 2, R)
    return {"qm": ea * eb % R, "ql": ea, "qr": eb, "qo": ec,
            "s3": (-alpha * a2 % R * beta % R * ezw) % R,
            "z": (alpha * a1 + alpha * alpha % R * l1) % R,
            "tlo": (-zh) % R, "tmid": (-zh * zn2) % R, "thi": (-zh * zn2 % R * zn2) % R,
            "r0": (pi - alpha * alpha % R * l1 - alpha * a2 % R * (ec + gamma) % R * ezw) % R}


# ---- verifier ----------------------------------------------------------------------------------------------
def verify(key, public, proof):
    com = key["com"]
    if any(not (0 <= int(x) < R) for x in public) or any(not (0 <= int(e) < R) for e in proof["ev"]):
        return False
    gamma = challenge("gamma", vk_digest(key), *public, proof["a"], proof["b"], proof["c"])
    beta = challenge("beta", gamma)
    alpha = challenge("alpha", beta, proof["z"])
    zeta = challenge("zeta", alpha, proof["tlo"], proof["tmid"], proof["thi"])
    ev = proof["ev"]
    v = challenge("v", zeta, *ev)
    # the folding challenge of the two openings is chained to the whole transcript through v (which
    # binds zeta, every evaluation and, through zeta / alpha / beta / gamma, all nine commitments and
    # the public inputs) -- as gnark derives its KZG folding randomness from digests, points and
    # claimed values; a u that depended on [W_zeta], [W_zeta_w] alone could be fixed in advance
    u = challenge("u", v, proof["wz"], proof["wzw"])
    sc = lin_scalars(key, public, beta, gamma, alpha, zeta, ev)
    w = root(key["log_n"])
    add, mul, neg = pyref.g1_add, pyref.g1_mul, pyref.g1_neg
    F = None
    for pt, s in ((com["qm"], sc["qm"]), (com["ql"], sc["ql"]), (com["qr"], sc["qr"]),
                  (com["qo"], sc["qo"]), (com["qc"], 1), (com["s3"], sc["s3"]), (proof["z"], sc["z"]),
                  (proof["tlo"], sc["tlo"]), (proof["tmid"], sc["tmid"]), (proof["thi"], sc["thi"])):
        if pt is not None:
            F = add(F, mul(pt, s))
    E = (-sc["r0"]) % R
    vp = 1
    for pt, e in ((proof["a"], ev[0]), (proof["b"], ev[1]), (proof["c"], ev[2]), (com["s1"], ev[3]),
                  (com["s2"], ev[4])):
        vp = vp * v % R
        if pt is not None:
            F = add(F, mul(pt, vp))
        E = (E + vp * e) % R
    F = add(F, mul(proof["z"], u)) if proof["z"] is not None else F
    E = (E + u * ev[5]) % R
    g1 = (1, 2)
    lhs = add(proof["wz"], mul(proof["wzw"], u) if proof["wzw"] is not None else None)
    rhs = add(add(mul(proof["wz"], zeta) if proof["wz"] is not None else None,
                  mul(proof["wzw"], u * zeta % R * w % R) if proof["wzw"] is not None else None),
              add(F, neg(mul(g1, E))))
    g2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))
    return pyref.pairing_product_is_one([(lhs, key["g2_tau"]), (neg(rhs), g2)])
