"""PLONK over BN254 with KZG commitments: host-side mirror of gnark's ``backend/plonk`` for this
framework (BASELINE config 5 names the PLONK backend) [UPSTREAM-RECALL; parity unpinned: the
reference's own test of that circuit proves with Groth16, ecc/secp256k1/ecdsa/address_test.go:57,
and holds no PLONK vector].

* ``setup``  -- plonk.Setup: SRS from a seeded tau (test setup, like gnark's ``test/unsafekzg``),
  selector / permutation polynomials from the sparse constraint system (frontend/scs.py), their
  coefficient forms, coset evaluations (GPU NTT) and commitments (GPU MSM).
* ``Prover`` -- plonk.Prove for a batch of independent witnesses: the five rounds run on the GPU
  (csrc/plonk.hip through zkmi_plonk_round1..5); between rounds the host derives the Fiat-Shamir
  challenges (SHA-256), as gnark does.  No CPU fallback.
* ``verify`` -- plonk.Verify on the host (pairing from verify.py).

Protocol: a, b, c blinded by (b1 X + b2) Z_H, z by (b7 X^2 + b8 X + b9) Z_H;
gamma = H("gamma", vk, public, [a], [b], [c]); beta = H("beta", gamma); alpha = H("alpha", beta,
[z]); zeta = H("zeta", alpha, [t_lo], [t_mid], [t_hi]); v = H("v", zeta, evaluations); u = H("u", v,
[W_zeta], [W_zeta_w]).  Quotient chunks hold n + 2 coefficients.
"""
from __future__ import annotations

import ctypes as C
import functools
import hashlib
import random

import numpy as np

from . import lib as _lib
from . import verify as _verify
from .frontend.compile import array_to_ints, ints_to_array, to_mont_array
from .frontend.scs import ScsCircuit
from .groth16 import g1_gen_mont

R = _verify.R
P = _verify.P
K = (1, 5, 25)
_G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634),
           (8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _inv(x):
    # extended Euclid in C (a few microseconds) rather than a 254-bit modular power (~80 us): the
    # transcript computes several inverses per proof between the GPU rounds
    x %= R
    return pow(x, -1, R) if x else 0


@functools.lru_cache(maxsize=None)
def root_of_unity(log_n):
    return pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)


def challenge(label, *parts):
    """SHA-256(label || parts) mod r: ints as 32 big-endian bytes, G1 points as x || y big-endian
    (infinity = 64 zero bytes)."""
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


class PlonkDesc(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("n_public", C.c_uint32), ("coef", C.c_void_p),
                ("coset", C.c_void_p), ("sigma", C.c_void_p), ("omega", C.c_void_p),
                ("coset_x", C.c_void_p), ("l1_coset", C.c_void_p), ("zh_inv", C.c_void_p),
                ("srs_g1", C.c_void_p), ("window_bits", C.c_uint32), ("max_batch", C.c_uint32),
                ("lag_g1", C.c_void_p * 3), ("lag_rows", C.c_void_p * 3), ("lag_k", C.c_uint32)]


class ProvingKey:
    """plonk.ProvingKey + VerifyingKey material (numpy, gnark's memory image)."""

    def __init__(self):
        self.log_n = 0
        self.n_public = 0
        self.coef = self.coset = self.sigma = self.omega = self.coset_x = self.l1 = None
        self.zh_inv = self.srs_g1 = None
        self.srs_lagrange = None   # n + 2 points: [L_i(tau)]_1, [tau^(n+1) - tau]_1, [tau^n - 1]_1
        self.com = {}              # name -> G1 point (ints) of qL..qC, S1..S3
        self.g2_tau = None
        self.vk_digest = b""


_NAMES = ("ql", "qr", "qo", "qm", "qc", "s1", "s2", "s3")


def setup(ctx: _lib.Context, scs: ScsCircuit, seed: int) -> ProvingKey:
    """Seeded (unsafe, test) SRS + preprocessed polynomials.  Heavy parts on the GPU: the SRS powers
    (zkmi_fixed_base_mul), the 8 inverse NTTs and 8 coset NTTs (zkmi_ntt_batch with the polynomials
    as the batch), the 8 commitments (zkmi_msm_batch)."""
    tau = random.Random(seed).randrange(2, R)
    log_n = scs.log_n
    n, m = 1 << log_n, 4 << log_n
    pk = ProvingKey()
    pk.log_n, pk.n_public = log_n, scs.n_public - 1
    pw, t = [], 1
    for _ in range(n + 6):
        pw.append(t)
        t = t * tau % R
    pk.srs_g1 = np.zeros((n + 6, 8), dtype=np.uint64)
    ctx.fixed_base_mul(1, g1_gen_mont(), to_mont_array(pw), n + 6, pk.srs_g1)
    w = root_of_unity(log_n)
    om, x = [], 1
    for _ in range(n):
        om.append(x)
        x = x * w % R
    # Lagrange-form SRS (gnark's plonk.Setup takes one beside the canonical SRS [UPSTREAM-RECALL]):
    # L_i(tau) = w^i (tau^n - 1) / (n (tau - w^i)), one batched inversion; the two points that carry
    # the blinding (b1 X + b2)(X^n - 1) of a wire polynomial committed in this basis follow
    den = [(tau - v) % R for v in om]
    pref, acc = [], 1
    for dv in den:
        pref.append(acc)
        acc = acc * dv % R
    inv = _inv(acc)
    zn = (pow(tau, n, R) - 1) * _inv(n) % R
    lag = [0] * n
    for i in range(n - 1, -1, -1):
        lag[i] = om[i] * zn % R * (inv * pref[i] % R) % R
        inv = inv * den[i] % R
    lag += [(pow(tau, n + 1, R) - tau) % R, (pow(tau, n, R) - 1) % R]
    pk.srs_lagrange = np.zeros((n + 2, 8), dtype=np.uint64)
    ctx.fixed_base_mul(1, g1_gen_mont(), to_mont_array(lag), n + 2, pk.srs_lagrange)
    ident = [k * v % R for k in K for v in om]
    sig = [[ident[int(scs.sigma[c * n + r])] for r in range(n)] for c in range(3)]
    lag = [scs.qL, scs.qR, scs.qO, scs.qM, scs.qC] + sig
    lag_arr = np.stack([to_mont_array([int(v) % R for v in col]) for col in lag])      # [8, n, 4]
    pk.sigma = np.ascontiguousarray(lag_arr[5:8])
    coef = lag_arr.copy()
    ctx.ntt_batch(coef, log_n, 8, inverse=True, coset=False)
    pk.coef = coef
    big = np.zeros((8, m, 4), dtype=np.uint64)
    big[:, :n] = coef
    ctx.ntt_batch(big, log_n + 2, 8, inverse=False, coset=True)
    pk.coset = big
    pk.omega = to_mont_array(om)
    w4 = root_of_unity(log_n + 2)
    xs, x = [], 5
    for _ in range(m):
        xs.append(x)
        x = x * w4 % R
    pk.coset_x = to_mont_array(xs)
    ninv = _inv(n)
    zh = [(pow(xs[j], n, R) - 1) % R for j in range(4)]
    pk.zh_inv = to_mont_array([_inv(v) for v in zh])
    # L1(x) = Z_H(x) / (n (x - 1)): batch inversion of the 4n denominators
    den = [(v - 1) % R for v in xs]
    pref, acc = [], 1
    for d in den:
        pref.append(acc)
        acc = acc * d % R
    inv = _inv(acc)
    l1 = [0] * m
    for j in range(m - 1, -1, -1):
        l1[j] = zh[j % 4] * ninv % R * (inv * pref[j] % R) % R
        inv = inv * den[j] % R
    pk.l1 = to_mont_array(l1)
    # commitments of the 8 key polynomials: one MSM with the polynomials as the batch
    h = ctx.msm_bases_load(1, pk.srs_g1[:n], n, 0)
    pts = np.zeros((8, 8), dtype=np.uint64)
    ctx.msm_batch(h, coef, 8, pts)
    ctx.msm_bases_free(h)
    pk.com = {name: _verify.g1_from_image(pts[i]) for i, name in enumerate(_NAMES)}
    # [tau]_2 on the host (one scalar multiplication; the verifier's side)
    pk.g2_tau = _g2_mul(_G2_GEN, tau)
    pk.vk_digest = challenge("vk", pk.log_n, pk.n_public,
                             *[pk.com[k] for k in _NAMES]).to_bytes(32, "big")
    return pk


# ---- small G2 arithmetic for [tau]_2 (host, once per key) -------------------------------------------
def _f2m(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _f2inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * d % P, -a[1] * d % P)


def _g2_add(p, q):
    if p is None or q is None:
        return p if q is None else q
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if ((y1[0] + y2[0]) % P, (y1[1] + y2[1]) % P) == (0, 0):
            return None
        lam = _f2m(_f2m((3, 0), _f2m(x1, x1)), _f2inv(((2 * y1[0]) % P, (2 * y1[1]) % P)))
    else:
        lam = _f2m(((y2[0] - y1[0]) % P, (y2[1] - y1[1]) % P),
                   _f2inv(((x2[0] - x1[0]) % P, (x2[1] - x1[1]) % P)))
    l2 = _f2m(lam, lam)
    x3 = ((l2[0] - x1[0] - x2[0]) % P, (l2[1] - x1[1] - x2[1]) % P)
    t = _f2m(lam, ((x1[0] - x3[0]) % P, (x1[1] - x3[1]) % P))
    return x3, ((t[0] - y1[0]) % P, (t[1] - y1[1]) % P)


def _g2_mul(p, k):
    out = None
    while k:
        if k & 1:
            out = _g2_add(out, p)
        p = _g2_add(p, p)
        k >>= 1
    return out


def lin_scalars(pk, public, beta, gamma, alpha, zeta, ev):
    """Per-proof scalars of the linearisation polynomial
    r(X) = qm qM + ql qL + qr qR + qo qO + qC + s3 S3 + z z(X) + tlo t_lo + tmid t_mid + thi t_hi + r0."""
    n = 1 << pk.log_n
    ea, eb, ec, es1, es2, ezw = ev
    zh = (pow(zeta, n, R) - 1) % R
    l1 = zh * _inv(n) % R * _inv(zeta - 1) % R
    w = root_of_unity(pk.log_n)
    pi = 0
    for j, x in enumerate(public):
        wj = pow(w, j, R)
        pi = (pi - x * (zh * wj % R * _inv(n) % R * _inv(zeta - wj) % R)) % R
    a1 = (ea + beta * zeta + gamma) * (eb + 5 * beta * zeta + gamma) % R * \
        (ec + 25 * beta * zeta + gamma) % R
    a2 = (ea + beta * es1 + gamma) * (eb + beta * es2 + gamma) % R
    zn2 = pow(zeta, n + 2, R)
    return {"qm": ea * eb % R, "ql": ea, "qr": eb, "qo": ec,
            "s3": (-alpha * a2 % R * beta % R * ezw) % R,
            "z": (alpha * a1 + alpha * alpha % R * l1) % R,
            "tlo": (-zh) % R, "tmid": (-zh * zn2) % R, "thi": (-zh * zn2 % R * zn2) % R,
            "r0": (pi - alpha * alpha % R * l1 - alpha * a2 % R * (ec + gamma) % R * ezw) % R}


class Proof:
    """plonk.Proof: 9 commitments (G1, ints or None) + 6 evaluations."""
    FIELDS = ("a", "b", "c", "z", "tlo", "tmid", "thi", "wz", "wzw")

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, kw.get(f))
        self.ev = tuple(kw.get("ev", ()))

    def __eq__(self, other):
        return all(getattr(self, f) == getattr(other, f) for f in self.FIELDS) and self.ev == other.ev


class Prover:
    """Device-resident (constraint system, PLONK key); ``prove`` runs the five GPU rounds."""

    def __init__(self, ctx: _lib.Context, scs: ScsCircuit, pk: ProvingKey, window_bits: int = 0,
                 max_batch: int = 64, lagrange=None):
        """lagrange: commit the wire columns a, b, c in the Lagrange basis (zkmi_plonk_pk_desc.lag_*)
        instead of their coefficient forms.  None = when at most 40 % of the rows of a column hold
        values the frontend cannot bound to one bit (Keccak, bit decompositions): the columns'
        values are the scalars then, and all-zero digits cost nothing in the subset-sum tables."""
        self.ctx, self.scs, self.pk = ctx, scs, pk
        if lagrange is None:
            lagrange = scs.n_nonbit_rows * 5 <= 2 * (1 << scs.log_n) and scs.log_n >= 10
        self.lagrange = bool(lagrange) and pk.srs_lagrange is not None
        self._consts = to_mont_array(scs.consts) if scs.consts else np.zeros((0, 4), np.uint64)
        prog = self._prog = np.ascontiguousarray(scs.vprogram, dtype=np.uint32)
        cd = _lib.CsDesc(scs.n_wires, scs.n_public, scs.n_secret, scs.n_constraints, scs.v_n_slots,
                         scs.v_n_rows, len(scs.consts), scs.lanes_per_proof, prog.ctypes.data,
                         self._consts.ctypes.data)
        self.cs_h = ctx.cs_load(cd)
        self._keep = [np.ascontiguousarray(x) for x in (pk.coef, pk.coset, pk.sigma, pk.omega,
                                                        pk.coset_x, pk.l1, pk.zh_inv, pk.srs_g1)]
        lag_g1 = (C.c_void_p * 3)()
        lag_rows = (C.c_void_p * 3)()
        lag_k = 0
        if self.lagrange:
            n = 1 << pk.log_n
            lag_k = 10
            for c in range(3):
                # blinding points first (their scalars are field elements), then the column's rows
                order = np.concatenate([np.array([n, n + 1], dtype=np.uint32), scs.lag_order[c]])
                pts = np.ascontiguousarray(pk.srs_lagrange[order])
                self._keep += [order, pts]
                lag_g1[c] = pts.ctypes.data
                lag_rows[c] = order.ctypes.data
        d = PlonkDesc(pk.log_n, pk.n_public, *[x.ctypes.data for x in self._keep[:8]], window_bits,
                      max_batch, lag_g1, lag_rows, lag_k)
        h = C.c_void_p()
        ctx._check(ctx.lib.zkmi_plonk_pk_load(ctx.h, C.byref(d), C.byref(h)), "zkmi_plonk_pk_load")
        self.pk_h = h

    def close(self):
        if getattr(self, "pk_h", None):
            self.ctx.lib.zkmi_plonk_pk_free(self.ctx.h, self.pk_h)
            self.pk_h = None
        if getattr(self, "cs_h", None):
            self.ctx.cs_free(self.cs_h)
            self.cs_h = None

    def prove_raw(self, inputs, blind):
        """plonk.Prove through zkmi_plonk_prove: the five rounds with the transcript hashed in C++
        inside the library (no Python between the rounds).  Returns (records uint64 [batch, 96]: nine
        G1 points then six evaluations, gnark's Montgomery image; status [batch]); ``proofs_of``
        turns records into Proof values."""
        batch = inputs.shape[0]
        if tuple(inputs.shape) != (batch, self.scs.n_inputs, 4) or tuple(blind.shape) != (batch, 9, 4):
            raise ValueError("inputs must be [batch, n_inputs, 4], blind [batch, 9, 4]")
        inputs, blind = np.ascontiguousarray(inputs), np.ascontiguousarray(blind)
        rec = np.zeros((batch, 96), dtype=np.uint64)
        status = np.zeros(batch, dtype=np.int32)
        vkd = (C.c_uint8 * 32).from_buffer_copy(self.pk.vk_digest)
        self.ctx._check(self.ctx.lib.zkmi_plonk_prove(self.ctx.h, self.pk_h, self.cs_h,
                                                      inputs.ctypes.data, batch, blind.ctypes.data,
                                                      vkd, rec.ctypes.data, status.ctypes.data),
                        "zkmi_plonk_prove")
        return rec, status

    @staticmethod
    def proofs_of(rec):
        """records of prove_raw -> list of Proof"""
        rinv = pow(1 << 256, R - 2, R)
        out = []
        for r in np.ascontiguousarray(rec, dtype=np.uint64).reshape(-1, 96):
            pts = [_verify.g1_from_image(r[8 * k:8 * k + 8]) for k in range(9)]
            ev = tuple(v * rinv % R for v in array_to_ints(r[72:96].reshape(6, 4)))
            out.append(Proof(**dict(zip(Proof.FIELDS, pts)), ev=ev))
        return out

    def prove(self, inputs, blind):
        """inputs: [batch, n_inputs, 4] Montgomery (public first); blind: [batch, 9, 4] Montgomery.
        Returns (list of Proof, status [batch]).  The five rounds one by one with the transcript in
        Python: the reference implementation of the transcript that ``prove_raw`` (C++) is tested
        against."""
        ctx, lib, pk = self.ctx, self.ctx.lib, self.pk
        batch = inputs.shape[0]
        if tuple(inputs.shape) != (batch, self.scs.n_inputs, 4) or tuple(blind.shape) != (batch, 9, 4):
            raise ValueError("inputs must be [batch, n_inputs, 4], blind [batch, 9, 4]")
        inputs, blind = np.ascontiguousarray(inputs), np.ascontiguousarray(blind)
        n_pub = pk.n_public
        rinv = pow(1 << 256, R - 2, R)
        pubs = [[v * rinv % R for v in array_to_ints(inputs[i, :n_pub])] for i in range(batch)]
        pts = lambda arr: [_verify.g1_from_image(arr[i]) for i in range(arr.shape[0])]
        status = np.zeros(batch, dtype=np.int32)
        c_abc = np.zeros((batch, 3, 8), dtype=np.uint64)
        ctx._check(lib.zkmi_plonk_round1(ctx.h, self.pk_h, self.cs_h, inputs.ctypes.data, batch,
                                         blind.ctypes.data, c_abc.ctypes.data, status.ctypes.data),
                   "zkmi_plonk_round1")
        A, B, Cc = (pts(c_abc[:, k]) for k in range(3))
        gamma = [challenge("gamma", pk.vk_digest, *pubs[i], A[i], B[i], Cc[i]) for i in range(batch)]
        beta = [challenge("beta", g) for g in gamma]
        bg = np.stack([to_mont_array([beta[i], gamma[i]]) for i in range(batch)])
        c_z = np.zeros((batch, 8), dtype=np.uint64)
        ctx._check(lib.zkmi_plonk_round2(ctx.h, self.pk_h, bg.ctypes.data, c_z.ctypes.data),
                   "zkmi_plonk_round2")
        Z = pts(c_z)
        alpha = [challenge("alpha", beta[i], Z[i]) for i in range(batch)]
        al = to_mont_array(alpha)
        c_t = np.zeros((batch, 3, 8), dtype=np.uint64)
        ctx._check(lib.zkmi_plonk_round3(ctx.h, self.pk_h, al.ctypes.data, c_t.ctypes.data),
                   "zkmi_plonk_round3")
        TL, TM, TH = (pts(c_t[:, k]) for k in range(3))
        zeta = [challenge("zeta", alpha[i], TL[i], TM[i], TH[i]) for i in range(batch)]
        w = root_of_unity(pk.log_n)
        zz = np.stack([to_mont_array([z, z * w % R]) for z in zeta])
        ev_arr = np.zeros((batch, 6, 4), dtype=np.uint64)
        ctx._check(lib.zkmi_plonk_round4(ctx.h, self.pk_h, zz.ctypes.data, ev_arr.ctypes.data),
                   "zkmi_plonk_round4")
        rinv = pow(1 << 256, R - 2, R)
        evs = [tuple(v * rinv % R for v in array_to_ints(ev_arr[i])) for i in range(batch)]
        rows = []
        for i in range(batch):
            v = challenge("v", zeta[i], *evs[i])
            sc = lin_scalars(pk, pubs[i], beta[i], gamma[i], alpha[i], zeta[i], evs[i])
            c0, vp = sc["r0"], 1
            for e in evs[i][:5]:
                vp = vp * v % R
                c0 = (c0 - vp * e) % R
            rows.append(to_mont_array([sc["qm"], sc["ql"], sc["qr"], sc["qo"], sc["s3"], sc["z"],
                                       sc["tlo"], sc["tmid"], sc["thi"], c0, v, zeta[i],
                                       zeta[i] * w % R, evs[i][5]]))
        sc_arr = np.stack(rows)
        c_w = np.zeros((batch, 2, 8), dtype=np.uint64)
        ctx._check(lib.zkmi_plonk_round5(ctx.h, self.pk_h, sc_arr.ctypes.data, c_w.ctypes.data),
                   "zkmi_plonk_round5")
        WZ, WZW = pts(c_w[:, 0]), pts(c_w[:, 1])
        proofs = [Proof(a=A[i], b=B[i], c=Cc[i], z=Z[i], tlo=TL[i], tmid=TM[i], thi=TH[i],
                        wz=WZ[i], wzw=WZW[i], ev=evs[i]) for i in range(batch)]
        return proofs, status


def verify(pk: ProvingKey, public, proof: Proof) -> bool:
    """plonk.Verify: recompute the challenges, assemble the linearisation commitment, one pairing
    check of the two batched KZG openings."""
    com = pk.com
    # canonical encodings only: challenge() reduces mod r, so an unreduced public input or claimed
    # evaluation would alias a reduced one
    if any(not (0 <= int(x) < R) for x in public) or any(not (0 <= int(e) < R) for e in proof.ev):
        return False
    gamma = challenge("gamma", pk.vk_digest, *public, proof.a, proof.b, proof.c)
    beta = challenge("beta", gamma)
    alpha = challenge("alpha", beta, proof.z)
    zeta = challenge("zeta", alpha, proof.tlo, proof.tmid, proof.thi)
    ev = proof.ev
    v = challenge("v", zeta, *ev)
    # the folding challenge of the two openings is chained to the whole transcript through v (which
    # binds zeta, every evaluation and, through zeta / alpha / beta / gamma, all nine commitments and
    # the public inputs) -- as gnark derives its KZG folding randomness from digests, points and
    # claimed values; a u that depended on [W_zeta], [W_zeta_w] alone could be fixed in advance
    u = challenge("u", v, proof.wz, proof.wzw)
    sc = lin_scalars(pk, public, beta, gamma, alpha, zeta, ev)
    w = root_of_unity(pk.log_n)
    add, mul, neg = _verify._g1_add, _verify._g1_mul, _verify._g1_neg
    sm = lambda pt, s: None if pt is None else mul(pt, s)
    F = None
    for pt, s in ((com["qm"], sc["qm"]), (com["ql"], sc["ql"]), (com["qr"], sc["qr"]),
                  (com["qo"], sc["qo"]), (com["qc"], 1), (com["s3"], sc["s3"]), (proof.z, sc["z"]),
                  (proof.tlo, sc["tlo"]), (proof.tmid, sc["tmid"]), (proof.thi, sc["thi"])):
        F = add(F, sm(pt, s))
    E = (-sc["r0"]) % R
    vp = 1
    for pt, e in ((proof.a, ev[0]), (proof.b, ev[1]), (proof.c, ev[2]), (com["s1"], ev[3]),
                  (com["s2"], ev[4])):
        vp = vp * v % R
        F = add(F, sm(pt, vp))
        E = (E + vp * e) % R
    F = add(F, sm(proof.z, u))
    E = (E + u * ev[5]) % R
    lhs = add(proof.wz, sm(proof.wzw, u))
    rhs = add(add(sm(proof.wz, zeta), sm(proof.wzw, u * zeta % R * w % R)),
              add(F, neg(mul((1, 2), E))))
    for pt in (proof.a, proof.b, proof.c, proof.z, proof.tlo, proof.tmid, proof.thi, proof.wz,
               proof.wzw):
        if not _verify._on_g1(pt):
            return False
    return _verify.pairing_product_is_one([(lhs, pk.g2_tau), (neg(rhs), _G2_GEN)])
