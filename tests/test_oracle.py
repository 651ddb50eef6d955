"""Pins the CPU oracles (oracle/pyref.py, oracle/c) before anything is compared against them.

The reference holds no prover-level vector ("parity unpinned"), so the chain of
evidence is: public constants K6 -> field arithmetic; naive DFT -> NTT; naive double-and-add ->
Pippenger; trapdoor closed form + pairing equation -> the complete Groth16 proof.
"""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import (array_to_ints, from_mont_array,
                                                          ints_to_array, to_mont_array)
from oracle import cref, pyref
from tests import helpers as H


def test_k6_constants():
    r, p = pyref.R, pyref.P
    assert r == 21888242871839275222246405745257275088548364400416034343698204186575808495617
    assert p == 21888242871839275222246405745257275088696311157297823662689037894645226208583
    assert (r - 1) % (1 << 28) == 0 and (r - 1) % (1 << 29) != 0
    w = pow(5, (r - 1) >> 28, r)
    assert w == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(5, (r - 1) // 2, r) == r - 1                      # 5 is a non-residue
    assert (1 << 256) % r == 6350874878119819312338956282401532410528162663560392320966563075034087161851
    assert (1 << 256) % p == 6350874878119819312338956282401532409788428879151445726012394534686998597021
    assert (-pow(r, -1, 1 << 64)) % (1 << 64) == 14042775128853446655
    assert (-pow(p, -1, 1 << 64)) % (1 << 64) == 9786893198990664585
    assert pyref.g1_on_curve(pyref.G1_GEN) and pyref.g1_mul(pyref.G1_GEN, r) is None
    assert pyref.g2_on_curve(pyref.G2_GEN) and pyref.g2_mul(pyref.G2_GEN, r) is None
    # the C oracle's Montgomery constants behave: to_mont(1) == R mod r
    one = cref.fr_to_mont(ints_to_array([1]))
    assert array_to_ints(one) == [(1 << 256) % r]


def test_c_field_ops_vs_python():
    rng = random.Random(2)
    xs = [0, 1, pyref.R - 1] + [rng.randrange(pyref.R) for _ in range(200)]
    ys = [pyref.R - 1, 0, pyref.R - 1] + [rng.randrange(pyref.R) for _ in range(200)]
    a, b = to_mont_array(xs), to_mont_array(ys)
    assert from_mont_array(cref.fr_mul(a, b)) == [x * y % pyref.R for x, y in zip(xs, ys)]
    assert from_mont_array(cref.fr_inv(a)) == [pow(x, pyref.R - 2, pyref.R) for x in xs]
    xq = [rng.randrange(pyref.P) for _ in range(100)]
    yq = [rng.randrange(pyref.P) for _ in range(100)]
    got = H.fq_unmont(cref.fq_mul(H.fq_mont(xq), H.fq_mont(yq)))
    assert got == [x * y % pyref.P for x, y in zip(xq, yq)]


@pytest.mark.parametrize("log_n", [1, 2, 5, 7])
def test_ntt_vs_naive_dft_and_python(log_n):
    rng = random.Random(log_n)
    n = 1 << log_n
    xs = [rng.randrange(pyref.R) for _ in range(n)]
    d = to_mont_array(xs)
    for inv in (0, 1):
        for coset in (0, 1):
            assert np.array_equal(cref.ntt(d, log_n, inv, coset), cref.dft_naive(d, log_n, inv, coset))
    w = pyref.root_of_unity(log_n)
    want = [sum(x * pow(5 * pow(w, k, pyref.R), i, pyref.R) for i, x in enumerate(xs)) % pyref.R
            for k in range(n)]
    assert from_mont_array(cref.ntt(d, log_n, 0, 1)) == want
    assert np.array_equal(cref.ntt(cref.ntt(d, log_n, 0, 1), log_n, 1, 1), d)


def test_compute_h_is_the_quotient():
    """h*Z == A*B - C as polynomials, checked at a random point."""
    rng = random.Random(5)
    log_n, n = 6, 64
    a, b = ([rng.randrange(pyref.R) for _ in range(n)] for _ in range(2))
    c = [x * y % pyref.R for x, y in zip(a, b)]          # satisfied on the whole domain
    h = from_mont_array(cref.compute_h(to_mont_array(a), to_mont_array(b), to_mont_array(c), log_n))
    assert h[n - 1] == 0                                   # deg h <= n - 2
    x = rng.randrange(pyref.R)
    lag = pyref.lagrange_at(x, log_n, n)
    ev = lambda v: sum(l * y for l, y in zip(lag, v)) % pyref.R
    hx = sum(coef * pow(x, i, pyref.R) for i, coef in enumerate(h)) % pyref.R
    assert hx * (pow(x, n, pyref.R) - 1) % pyref.R == (ev(a) * ev(b) - ev(c)) % pyref.R


@pytest.mark.parametrize("group", [1, 2])
def test_msm_pippenger_vs_naive_vs_python(group):
    rng = random.Random(group)
    n = 30
    gen = H.g1_gen_mont() if group == 1 else H.g2_gen_mont()
    ks = [rng.randrange(1, pyref.R) for _ in range(n)]
    bases = cref.batch_mul(group, gen, to_mont_array(ks))
    sc = [rng.randrange(pyref.R) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, pyref.R - 1
    s = to_mont_array(sc)
    naive = cref.msm(group, bases, s, naive=True)
    for c in (3, 8, 13, 16):
        assert np.array_equal(cref.msm(group, bases, s, c=c), naive)
    total = sum(k * x for k, x in zip(ks, sc)) % pyref.R
    if group == 1:
        want = pyref.g1_mul(pyref.G1_GEN, total)
        assert tuple(H.fq_unmont(naive.reshape(-1, 4))) == want
    else:
        want = pyref.g2_mul(pyref.G2_GEN, total)
        v = H.fq_unmont(naive.reshape(-1, 4))
        assert ((v[0], v[1]), (v[2], v[3])) == want


def test_pairing_bilinear_and_nondegenerate():
    p2, q3 = pyref.g1_mul(pyref.G1_GEN, 2), pyref.g2_mul(pyref.G2_GEN, 3)
    p6 = pyref.g1_mul(pyref.G1_GEN, 6)
    assert pyref.pairing_product_is_one([(p2, q3), (pyref.g1_neg(p6), pyref.G2_GEN)])
    assert not pyref.pairing_product_is_one([(p2, q3), (pyref.g1_neg(p2), pyref.G2_GEN)])


def _pt(proof):
    g1 = lambda a: (lambda v: None if not any(v) else (v[0], v[1]))(H.fq_unmont(a.reshape(-1, 4)))
    g2 = lambda a: (lambda v: None if not any(v) else ((v[0], v[1]), (v[2], v[3])))(
        H.fq_unmont(a.reshape(-1, 4)))
    return g1(proof[0:8]), g1(proof[8:16]), g2(proof[16:32])


def test_groth16_c_oracle_vs_closed_form_and_pairing():
    """Config 1 end to end on the CPU: compile single-Poseidon circuit -> setup -> solve -> prove;
    the C oracle's proof equals the NTT/MSM-free closed form and satisfies the pairing equation."""
    cc = compile_circuit(circuits.PoseidonCircuit())
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    pk, vk, td = groth16.setup(cc, 7, mul)
    data = 297262668938251460872476410954775437897592223497      # poseidon_test.go:39
    inp = cc.assignment_vector({"Data": data, "Hash": pyref.poseidon_hash([data])})
    wires, *_ = cc.run_program(inp)
    r, s = 123456789, 987654321
    rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
    rc, proof = cref.groth16_prove(rh, ph, to_mont_array(inp), to_mont_array([r, s]))
    assert rc == 0
    got = _pt(proof)
    assert got == pyref.expected_proof(cc.constraints, cc.n_wires, cc.n_public, wires, td,
                                       pk.log_n, r, s)
    d = np.zeros(8, np.uint64)
    vkd = dict(alpha=_pt(np.concatenate([vk.g1_alpha, d, vk.g2_beta]))[0],
               beta=_pt(np.concatenate([d, d, vk.g2_beta]))[2],
               gamma=_pt(np.concatenate([d, d, vk.g2_gamma]))[2],
               delta=_pt(np.concatenate([d, d, vk.g2_delta]))[2],
               k=[_pt(np.concatenate([k, d, vk.g2_beta]))[0] for k in vk.g1_k])
    assert pyref.verify(vkd, wires[:cc.n_public], got)
    assert not pyref.verify(vkd, [1, (wires[1] + 1) % pyref.R], got)
    # different window sizes of the oracle's Pippenger give the same proof
    rc, proof2 = cref.groth16_prove(rh, ph, to_mont_array(inp), to_mont_array([r, s]), msm_c=4)
    assert rc == 0 and np.array_equal(proof, proof2)


def _regression_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "groth16_regression.json")
    fx = json.load(open(path))
    for name, circuit in (("poseidon", circuits.PoseidonCircuit()),
                          ("smt6", circuits.smt_inclusion_circuit(6))):
        yield name, circuit, fx[name]


def test_groth16_regression_fixture_oracle():
    """Committed proof bytes (tests/golden/groth16_regression.json): same compile, same seeded
    setup, same (r, s) -> byte-identical proofs from the C oracle."""
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    for name, circuit, fx in _regression_cases():
        cc = compile_circuit(circuit)
        assert cc.fingerprint() == fx["fingerprint"], "constraint system changed: regenerate"
        pk, _, _ = groth16.setup(cc, fx["setup_seed"], mul)
        inp = np.stack([to_mont_array([int(x) for x in v]) for v in fx["inputs"]])
        rs = np.stack([to_mont_array([int(x) for x in v]) for v in fx["rs"]])
        proofs, status, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
        assert not status.any()
        assert [p.tobytes().hex() for p in proofs] == fx["proofs_hex"]
