"""Parity at BASELINE.json's full sizes (config 2: n = 2^16, 65 535-term MSMs, Arbo-160 circuit),
through size-independent properties plus spot checks against the C oracle."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
from gnark_crypto_primitives_amd.tree import smt_witness
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _rand_fr_array(rng, shape):
    """uniform 254-bit values reduced below r, as Montgomery-agnostic raw limbs"""
    a = rng.integers(0, 1 << 63, size=shape + (4,), dtype=np.uint64) * 2 + \
        rng.integers(0, 2, size=shape + (4,), dtype=np.uint64)
    a[..., 3] &= np.uint64(0x0fffffffffffffff)        # < 2^252 < r
    return a


def test_ntt_full_size_roundtrip_and_scaling(zk_ctx):
    from oracle import cref
    rng = np.random.default_rng(16)
    log_n, batch = 16, 64
    x = _rand_fr_array(rng, (batch, 1 << log_n))
    for coset in (0, 1):
        y = x.copy()
        zk_ctx.ntt_batch(y, log_n, batch, False, coset)
        assert not np.array_equal(y, x)
        # spot check one proof against the oracle's DIF/DIT implementation
        assert np.array_equal(y[3], cref.ntt(x[3], log_n, False, coset))
        z = y.copy()
        zk_ctx.ntt_batch(z, log_n, batch, True, coset)
        assert np.array_equal(z, x)                      # inverse(forward(x)) == x
    # linearity: NTT(2x) == 2 NTT(x)
    two = np.tile(H.to_mont_array([2])[0], (batch * (1 << log_n), 1))
    x2 = cref.fr_mul(x.reshape(-1, 4), two).reshape(x.shape)
    y, y2 = x.copy(), x2.copy()
    zk_ctx.ntt_batch(y, log_n, batch, False, 0)
    zk_ctx.ntt_batch(y2, log_n, batch, False, 0)
    assert np.array_equal(y2, cref.fr_mul(y.reshape(-1, 4), two).reshape(x.shape))


@pytest.mark.parametrize("group,n", [(1, 65535), (2, 27059)])
def test_msm_full_size_properties(zk_ctx, group, n):
    """unit vectors pick bases, MSM(s) + MSM(t) == MSM(s + t), and one vector against Pippenger."""
    from oracle import cref
    r = random.Random(group)
    rng = np.random.default_rng(group)
    gen = H.g1_gen_mont() if group == 1 else H.g2_gen_mont()
    ks = _rand_fr_array(rng, (n,))
    bases = np.zeros((n, 8 if group == 1 else 16), dtype=np.uint64)
    zk_ctx.fixed_base_mul(group, gen, ks, n, bases)
    assert np.array_equal(bases[:50], cref.batch_mul(group, gen, ks[:50]))
    h = zk_ctx.msm_bases_load(group, bases, n, 8)
    batch = 6
    sc = _rand_fr_array(rng, (batch, n))
    sc[0] = 0
    i0 = r.randrange(n)
    sc[0, i0] = H.to_mont_array([1])[0]                 # unit vector
    sc[1, 100:] = 0                                      # short support
    # row 4 = row 2 + row 3 (field addition done by the oracle)
    import ctypes as C
    a, b, out = sc[2].copy(), sc[3].copy(), np.zeros_like(sc[2])
    cref.lib().zkref_fr_add(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                            out.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    sc[4] = out
    res = np.zeros((batch, bases.shape[1]), dtype=np.uint64)
    zk_ctx.msm_batch(h, sc, batch, res)
    zk_ctx.msm_bases_free(h)
    assert np.array_equal(res[0], bases[i0])
    assert np.array_equal(res[1], cref.msm(group, bases[:100], sc[1, :100]))
    assert np.array_equal(res[4], cref.point_add(group, res[2], res[3]))
    assert np.array_equal(res[5], cref.msm(group, bases, sc[5], c=13))


def test_arbo160_prove_vs_oracle(zk_ctx):
    """The headline circuit itself (160 levels, 40 361 constraints, domain 2^16), 5 proofs,
    bit-exact against the C oracle; narrower windows than the bench to keep the tables small."""
    from oracle import cref
    cc = H.compiled("arbo160")
    assert cc.n_constraints == 40361 and cc.domain_log2() == 16
    pk, vk, td = groth16.setup(cc, 2, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, 6, 5)
    rng = random.Random(160)
    ws = [smt_witness.synthetic_inclusion(rng, 160, k) for k in (0, 1, 10, 40, 159)]
    inp = np.stack([to_mont_array(cc.assignment_vector(w)) for w in ws])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in ws])
    proofs, status = prover.prove(inp, rs)
    prover.close()
    assert not status.any()
    want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs, 8)
    assert not wstatus.any()
    assert np.array_equal(proofs, want)
