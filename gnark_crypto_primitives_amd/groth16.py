"""Groth16 over BN254: host-side mirror of gnark's ``backend/groth16`` for this framework.

* ``setup``   -- groth16.Setup (gnark backend/groth16/bn254/setup.go [UPSTREAM-RECALL]): trapdoor
  sampled from a seed, QAP evaluated at tau in Python ints, group elements produced by the GPU
  fixed-base kernel (``zkmi_fixed_base_mul``) or any ``mul(group, scalars) -> points`` callable.
  Key layout follows gnark's ProvingKey: G1{Alpha,Beta,Delta,A,B,Z,K}, G2{Beta,Delta,B}, with
  the InfinityA/B bitmaps expressed as the wire index of every retained point.
* ``Prover``  -- groth16.Prove for a *batch* of independent witnesses on one GPU: uploads the key
  once (zkmi_pk_load builds the HBM window tables), the constraint system once (zkmi_cs_load),
  then every ``prove`` call is one zkmi_prove_batch.  No CPU fallback.

Reference call sites this replaces: the ``test.Assert`` prover checks, e.g.
hash/emulated/bn254/poseidon/poseidon_test.go:72 (ProverSucceeded) and, under gnark's
``prover_checks`` tag, every SolvingSucceeded site such as tree/test/verifier_bn254_test.go:67.
"""
from __future__ import annotations

import ctypes as C
import random

import numpy as np

from . import lib as _lib
from .frontend.compile import CompiledCircuit, R, ints_to_array, to_mont_array

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
_MONT_P = (1 << 256) % P

G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def _fq_mont(xs):
    return ints_to_array([x % P * _MONT_P % P for x in xs])


def g1_gen_mont():
    return _fq_mont(G1_GEN).reshape(-1)


def g2_gen_mont():
    return _fq_mont([G2_GEN[0][0], G2_GEN[0][1], G2_GEN[1][0], G2_GEN[1][1]]).reshape(-1)


def _inv(x):
    # extended Euclid in C (a few microseconds) instead of a 254-bit modular power; 0 -> 0
    x %= R
    return pow(x, -1, R) if x else 0


def root_of_unity(log_n):
    return pow(pow(5, (R - 1) >> 28, R), 1 << (28 - log_n), R)


class ProvingKey:
    """gnark groth16 (bn254) ProvingKey, numpy arrays in gnark's memory layout (Montgomery)."""

    def __init__(self):
        self.log_n = 0
        self.n_wires = 0
        self.a_wire = self.b_wire = self.k_wire = None     # uint32
        self.g1_a = self.g1_b = self.g1_k = self.g1_z = self.g2_b = None
        self.g1_alpha = self.g1_beta = self.g1_delta = self.g2_beta = self.g2_delta = None
        # gnark ProvingKey.CommitmentKeys (pedersen.ProvingKey {Basis, BasisExpSigma}) plus the wire
        # lists of constraint.Groth16Commitments: [{"basis", "basis_exp_sigma" (n x 8 uint64),
        # "private", "hashed" (wire indices), "wire"}]
        self.commitment_keys = []

    def nbytes(self):
        return sum(x.nbytes for x in (self.g1_a, self.g1_b, self.g1_k, self.g1_z, self.g2_b))

    def infinity_maps(self):
        """gnark's ProvingKey.InfinityA / InfinityB ([]bool over all wires: True = the point is at
        infinity and not stored), as uint8 arrays."""
        inf_a = np.ones(self.n_wires, dtype=np.uint8)
        inf_b = np.ones(self.n_wires, dtype=np.uint8)
        inf_a[self.a_wire] = 0
        inf_b[self.b_wire] = 0
        return inf_a, inf_b


class VerifyingKey:
    def __init__(self):
        self.g1_alpha = self.g2_beta = self.g2_gamma = self.g2_delta = None
        self.g1_k = None            # one point per public wire (ONE first), then one per commitment
        # gnark VerifyingKey.CommitmentKey (pedersen.VerifyingKey {G, GSigmaNeg}) and
        # PublicAndCommitmentCommitted (here: wire indices hashed with each commitment)
        self.commitment_g = self.commitment_g_sigma_neg = None
        self.commitment_hashed = []
        self.commitment_wires = []


def sample_trapdoor(seed):
    rng = random.Random(seed)
    return tuple(rng.randrange(1, R) for _ in range(5))   # tau, alpha, beta, gamma, delta


def setup(cc: CompiledCircuit, seed, mul):
    """(pk, vk, trapdoor).  ``mul(group, scalars_mont[n,4]) -> points`` multiplies the group
    generator by each scalar (gnark: curve.BatchScalarMultiplicationG1/G2)."""
    tau, alpha, beta, gamma, delta = trapdoor = sample_trapdoor(seed)
    log_n = cc.domain_log2()
    n = 1 << log_n
    nw = cc.n_wires
    # Lagrange basis at tau for the constraint rows in use
    w = root_of_unity(log_n)
    zt = (pow(tau, n, R) - 1) % R
    if zt == 0:
        raise ValueError("tau lies in the evaluation domain")
    ninv = _inv(n)
    A, B, Cc = [0] * nw, [0] * nw, [0] * nw
    wk = 1
    for con in cc.constraints:
        lk = zt * wk % R * ninv % R * _inv(tau - wk) % R
        for vec, lc in ((A, con[0]), (B, con[1]), (Cc, con[2])):
            for wi, c in lc.items():
                vec[wi] = (vec[wi] + c * lk) % R
        wk = wk * w % R
    dinv, ginv = _inv(delta), _inv(gamma)
    pk, vk = ProvingKey(), VerifyingKey()
    pk.log_n, pk.n_wires = log_n, nw
    a_wire = [i for i in range(nw) if A[i]]
    b_wire = [i for i in range(nw) if B[i]]
    # commitment extension (gnark setup.go): private committed wires and commitment wires leave
    # pk.G1.K; the former become the Pedersen bases (their K scalar over gamma, like a public
    # wire's), the latter join vk.G1.K after the public wires
    coms = getattr(cc, "commitments", [])
    taken = set()
    for c in coms:
        taken.update(c["private"])
        taken.add(c["wire"])
    k_wire = [i for i in range(cc.n_public, nw) if i not in taken]
    pk.a_wire = np.array(a_wire, dtype=np.uint32)
    pk.b_wire = np.array(b_wire, dtype=np.uint32)
    pk.k_wire = np.array(k_wire, dtype=np.uint32)
    kk = [(beta * A[i] + alpha * B[i] + Cc[i]) % R for i in range(nw)]
    z = []
    t = zt * dinv % R
    for _ in range(n - 1):
        z.append(t)
        t = t * tau % R
    g1_scalars = ([A[i] for i in a_wire] + [B[i] for i in b_wire] +
                  [kk[i] * dinv % R for i in k_wire] + z +
                  [kk[i] * ginv % R for i in range(cc.n_public)] +
                  [kk[c["wire"]] * ginv % R for c in coms] + [alpha, beta, delta])
    if coms:
        rng = random.Random((int(seed) << 8) ^ 0xC0)
        sigma, rho = rng.randrange(1, R), rng.randrange(1, R)
        for c in coms:
            g1_scalars += [kk[i] * ginv % R for i in c["private"]]
            g1_scalars += [kk[i] * ginv % R * sigma % R for i in c["private"]]
    g1_pts = mul(1, to_mont_array(g1_scalars))
    o = 0

    def take(cnt):
        nonlocal o
        r = np.ascontiguousarray(g1_pts[o:o + cnt])
        o += cnt
        return r
    pk.g1_a, pk.g1_b, pk.g1_k, pk.g1_z = (take(len(a_wire)), take(len(b_wire)), take(len(k_wire)),
                                          take(n - 1))
    vk.g1_k = take(cc.n_public + len(coms))
    pk.g1_alpha, pk.g1_beta, pk.g1_delta = (take(1).reshape(-1) for _ in range(3))
    for c in coms:
        n_p = len(c["private"])
        pk.commitment_keys.append({"basis": take(n_p), "basis_exp_sigma": take(n_p),
                                   "private": list(c["private"]), "hashed": list(c["hashed"]),
                                   "wire": c["wire"]})
        vk.commitment_hashed.append(list(c["hashed"]))
        vk.commitment_wires.append(c["wire"])
    vk.g1_alpha = pk.g1_alpha
    g2_pts = mul(2, to_mont_array([B[i] for i in b_wire] + [beta, delta, gamma] +
                                  ([rho, (R - sigma) * rho % R] if coms else [])))
    nb = len(b_wire)
    if coms:
        vk.commitment_g, vk.commitment_g_sigma_neg = g2_pts[nb + 3].copy(), g2_pts[nb + 4].copy()
    pk.g2_b = np.ascontiguousarray(g2_pts[:nb])
    pk.g2_beta, pk.g2_delta = g2_pts[nb].copy(), g2_pts[nb + 1].copy()
    vk.g2_beta, vk.g2_delta, vk.g2_gamma = pk.g2_beta, pk.g2_delta, g2_pts[nb + 2].copy()
    return pk, vk, trapdoor


def commit_fn(pk: ProvingKey):
    """Host evaluation of the commitment hint for CompiledCircuit.run_program / run_vprogram (the
    CPU interpreters of the frontend tests): Pedersen commitment over pk.CommitmentKeys[i].Basis in
    Python integers, then hash_to_field -- the same value the GPU prover writes into the commitment
    wire (csrc/commit.hip)."""
    from . import hash_to_field, verify as _v
    bases = [[_v.g1_from_image(b) for b in ck["basis"]] for ck in pk.commitment_keys]

    def fn(idx, hashed, committed):
        acc = None
        for pt, s in zip(bases[idx], committed):
            if s % R and pt is not None:
                acc = _v._g1_add(acc, _v._g1_mul(pt, s % R))
        return hash_to_field.commitment_challenge(acc, hashed)
    return fn


def gpu_mul(ctx: _lib.Context):
    """``mul`` callable for ``setup`` backed by zkmi_fixed_base_mul."""
    def mul(group, scalars):
        n = scalars.shape[0]
        out = np.zeros((n, 8 if group == 1 else 16), dtype=np.uint64)
        if n:
            ctx.fixed_base_mul(group, g1_gen_mont() if group == 1 else g2_gen_mont(), scalars, n,
                               out)
        return out
    return mul


class Prover:
    """Device-resident (constraint system, proving key) pair; ``prove`` = one zkmi_prove_batch."""

    def __init__(self, ctx: _lib.Context, cc: CompiledCircuit, pk: ProvingKey,
                 window_bits_g1: int = 0, window_bits_g2: int = 0, *, max_batch: int = 0,
                 table_budget_bytes: int = 0, msm_chunk_factor: int = 0,
                 gnark_key_layout: bool = False, sparse_witness=None):
        """max_batch / table_budget_bytes / msm_chunk_factor: zkmi_pk_desc plan fields
        (0 = default).  sparse_witness: zkmi_pk_desc.sparse_witness; None = from the circuit (more
        than half of its wires are known booleans: Keccak, bit decompositions).  gnark_key_layout: describe the key the way
        gnark's ProvingKey does (InfinityA / InfinityB byte maps + nbPublic) instead of wire-index
        arrays; the loaded key is the same."""
        self.ctx, self.cc, self.pk = ctx, cc, pk
        self.n_inputs = cc.n_inputs
        self._consts = to_mont_array(cc.consts) if cc.consts else np.zeros((0, 4), np.uint64)
        prog = self._prog = np.ascontiguousarray(cc.vprogram, dtype=np.uint32)
        cd = _lib.CsDesc(cc.n_wires, cc.n_public, cc.n_secret, cc.n_constraints, cc.v_n_slots,
                         cc.v_n_rows, len(cc.consts), cc.lanes_per_proof, prog.ctypes.data,
                         self._consts.ctypes.data)
        self.cs_h = ctx.cs_load(cd)
        self._keep = [np.ascontiguousarray(x) for x in
                      (pk.a_wire, pk.b_wire, pk.k_wire, pk.g1_a, pk.g1_b, pk.g1_k, pk.g1_z, pk.g2_b,
                       pk.g1_alpha, pk.g1_beta, pk.g1_delta, pk.g2_beta, pk.g2_delta)]
        k = self._keep
        ptrs = [x.ctypes.data for x in k]
        inf_a = inf_b = None
        n_public = 0
        if gnark_key_layout:
            inf_a, inf_b = pk.infinity_maps()
            self._keep += [inf_a, inf_b]
            ptrs[0] = ptrs[1] = ptrs[2] = None
            n_public = cc.n_public
        # commitment extension: pk.CommitmentKeys + the wire lists of constraint.Groth16Commitments
        self.n_commitments = len(pk.commitment_keys)
        cdescs = (_lib.CommitmentDesc * max(self.n_commitments, 1))()
        for i, ck in enumerate(pk.commitment_keys):
            arrs = [np.ascontiguousarray(ck["private"], dtype=np.uint32),
                    np.ascontiguousarray(ck["hashed"], dtype=np.uint32),
                    np.ascontiguousarray(ck["basis"], dtype=np.uint64),
                    np.ascontiguousarray(ck["basis_exp_sigma"], dtype=np.uint64)]
            self._keep += arrs
            cdescs[i] = _lib.CommitmentDesc(len(ck["private"]), len(ck["hashed"]), ck["wire"], 0,
                                            *[a.ctypes.data for a in arrs])
        self._keep.append(cdescs)
        if len(getattr(cc, "commitments", [])) != self.n_commitments:
            raise ValueError("circuit and proving key disagree on the number of commitments")
        pd = _lib.PkDesc(pk.log_n, pk.n_wires, len(pk.a_wire), len(pk.b_wire), len(pk.k_wire),
                         pk.g1_z.shape[0], *ptrs, window_bits_g1, window_bits_g2,
                         inf_a.ctypes.data if inf_a is not None else None,
                         inf_b.ctypes.data if inf_b is not None else None, n_public,
                         max_batch, table_budget_bytes, cc.v_n_slots, msm_chunk_factor,
                         (2 if cc.n_boolean_wires * 100 >= cc.n_wires * 99 else
                          int(cc.n_boolean_wires * 2 > cc.n_wires)) if sparse_witness is None
                         else int(sparse_witness),
                         self.n_commitments,
                         C.addressof(cdescs) if self.n_commitments else None)
        self.pk_h = ctx.pk_load(pd)

    def load_r1cs(self):
        """zkmi_r1cs_load of the circuit's L, R, O (gnark: constraint.R1CS terms {CID, VID} and
        cs.Coefficients): afterwards ``submit_witness`` ships wire vectors only."""
        if getattr(self, "r1cs_h", None):
            return self.r1cs_h
        cc = self.cc
        coeffs = to_mont_array(cc.consts)
        keep = [coeffs]
        fields = []
        for ptr, col, cid in (cc.L, cc.Rm, cc.O):
            terms = np.ascontiguousarray(np.stack([cid, col], axis=1).astype(np.uint32))
            ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
            keep += [ptr, terms]
            fields += [ptr.ctypes.data, terms.ctypes.data]
        rd = _lib.R1csDesc(cc.n_wires, cc.n_constraints, len(cc.consts), coeffs.ctypes.data, *fields)
        self.r1cs_h = self.ctx.r1cs_load(rd)
        return self.r1cs_h

    def close(self):
        if getattr(self, "r1cs_h", None):
            self.ctx.r1cs_free(self.r1cs_h)
            self.r1cs_h = None
        if getattr(self, "pk_h", None):
            self.ctx.pk_free(self.pk_h)
            self.pk_h = None
        if getattr(self, "cs_h", None):
            self.ctx.cs_free(self.cs_h)
            self.cs_h = None

    def solve(self, inputs, want_wires=True, want_abc=False):
        """Witness solve only (cs.R1CS.Solve).  inputs: [batch, n_inputs, 4] Montgomery."""
        batch = inputs.shape[0]
        self._check_batch(inputs, (batch, self.n_inputs, 4), "inputs")
        wires = np.zeros((batch, self.cc.n_wires, 4), np.uint64) if want_wires else None
        abc = np.zeros((3, batch, self.cc.n_constraints, 4), np.uint64) if want_abc else None
        status = self.ctx.solve_batch(self.cs_h, np.ascontiguousarray(inputs), batch, wires, abc)
        return status, wires, abc

    def prove(self, inputs, rs, proofs_out=None, status_out=None):
        """inputs: [batch, n_inputs, 4] (numpy) or a device tensor; rs: [batch, 2, 4].
        Returns (proofs [batch, 32] uint64: Ar | Krs | Bs, status [batch] int32)."""
        batch = inputs.shape[0]
        self._check_batch(inputs, (batch, self.n_inputs, 4), "inputs")
        self._check_batch(rs, (batch, 2, 4), "rs")
        if self.n_commitments:          # keys with commitments: the pipelined pair, three outputs
            self.submit(inputs, rs)
            return self.collect(proofs_out, status_out)
        if proofs_out is None:
            proofs_out = np.zeros((batch, 32), dtype=np.uint64)
        if status_out is None:
            status_out = np.zeros(batch, dtype=np.int32)
        self.ctx.prove_batch(self.pk_h, self.cs_h, inputs, batch, rs, proofs_out, status_out)
        return proofs_out, status_out

    def prove_witness(self, wires, a, b, c, rs, proofs_out=None):
        """groth16.Prove from solved witnesses (zkmi_prove_witness_batch): ``wires`` [batch,
        n_wires, 4] full wire vectors (ONE first), ``a``/``b``/``c`` [batch, n_constraints, 4]
        constraint evaluations, all in gnark's Montgomery image -- what gnark's own solver yields."""
        batch = wires.shape[0]
        self._check_batch(wires, (batch, self.cc.n_wires, 4), "wires")
        for name, x in (("a", a), ("b", b), ("c", c)):
            self._check_batch(x, (batch, self.cc.n_constraints, 4), name)
        self._check_batch(rs, (batch, 2, 4), "rs")
        if proofs_out is None:
            proofs_out = np.zeros((batch, 32), dtype=np.uint64)
        self.ctx.prove_witness_batch(self.pk_h, wires, a, b, c, self.cc.n_constraints, batch, rs,
                                     proofs_out)
        return proofs_out

    def submit_witness(self, wires, rs, a=None, b=None, c=None):
        """Stage 1 of a prove from solved witnesses (zkmi_prove_witness_submit); ``collect`` returns
        the proofs.  Without a, b, c the R1CS matrices are loaded (once) and the device forms them
        from the wire vectors and checks a.b = c per proof."""
        batch = wires.shape[0]
        self._check_batch(wires, (batch, self.cc.n_wires, 4), "wires")
        self._check_batch(rs, (batch, 2, 4), "rs")
        if a is None:
            if b is not None or c is not None:
                raise ValueError("a, b, c: all three or none")
            self.ctx.prove_witness_submit(self.pk_h, self.load_r1cs(), wires, None, None, None, 0,
                                          batch, rs)
        else:
            for name, x in (("a", a), ("b", b), ("c", c)):
                self._check_batch(x, (batch, self.cc.n_constraints, 4), name)
            self.ctx.prove_witness_submit(self.pk_h, None, wires, a, b, c, self.cc.n_constraints,
                                          batch, rs)
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append(((wires, a, b, c), rs, batch))      # keep buffers alive

    @staticmethod
    def _check_batch(x, shape, name):
        """The C ABI takes raw pointers: refuse anything that is not the documented layout."""
        if tuple(x.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(x.shape)}")
        if isinstance(x, np.ndarray):
            if x.dtype != np.uint64 or not x.flags["C_CONTIGUOUS"]:
                raise ValueError(f"{name}: expected a C-contiguous uint64 array")
        elif hasattr(x, "is_contiguous"):
            if not x.is_contiguous() or x.element_size() != 8:
                raise ValueError(f"{name}: expected a contiguous 64-bit tensor")

    def submit(self, inputs, rs):
        """Stage 1 (inputs + witness solve) of a batch; overlaps the previous batch's stage 2."""
        self._check_batch(inputs, (inputs.shape[0], self.n_inputs, 4), "inputs")
        self._check_batch(rs, (inputs.shape[0], 2, 4), "rs")
        self.ctx.prove_submit(self.pk_h, self.cs_h, inputs, inputs.shape[0], rs)
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append((inputs, rs, inputs.shape[0]))     # keep buffers alive

    def collect(self, proofs_out=None, status_out=None, commitments_out=None):
        """Stage 2 (quotient, MSMs, assembly) of the oldest submitted batch.  Keys with the
        commitment extension return (proofs, status, commitments): commitments[p] = uint64
        [n_commitments + 1, 8]: proof.Commitments then proof.CommitmentPok."""
        _, _, batch = self._inflight.pop(0)
        if proofs_out is None:
            proofs_out = np.zeros((batch, 32), dtype=np.uint64)
        if status_out is None:
            status_out = np.zeros(batch, dtype=np.int32)
        if self.n_commitments:
            if commitments_out is None:
                commitments_out = np.zeros((batch, self.n_commitments + 1, 8), dtype=np.uint64)
            self.ctx.prove_collect(proofs_out, status_out, commitments_out)
            return proofs_out, status_out, commitments_out
        self.ctx.prove_collect(proofs_out, status_out)
        return proofs_out, status_out

    def prove_stream(self, batches):
        """Generator over (inputs, rs) batches, software-pipelined two deep."""
        it = iter(batches)
        try:
            self.submit(*next(it))
        except StopIteration:
            return
        for nxt in it:
            self.submit(*nxt)
            yield self.collect()
        yield self.collect()
