"""GPU parity of witness solve and Groth16 prove (through the C-ABI) against the oracles.

Bar: bit-exact.  Checker 1 = C oracle (gnark-style constraint-by-constraint solver, DIF/DIT NTT,
Pippenger MSM); checker 2 = closed-form proof from the trapdoor (no NTT/MSM at all); checker 3 =
pairing verification.  Prover-level results are pinned by nothing in the reference
("parity unpinned"); see DESIGN.md §Oracle.
"""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import circuits, groth16
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.frontend.compile import from_mont_array, to_mont_array
from gnark_crypto_primitives_amd.tree import smt_witness
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _pts(proof):
    def g1(a):
        v = H.fq_unmont(a.reshape(-1, 4))
        return None if not any(v) else (v[0], v[1])

    def g2(a):
        v = H.fq_unmont(a.reshape(-1, 4))
        return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))
    return g1(proof[0:8]), g1(proof[8:16]), g2(proof[16:32])


@pytest.fixture(scope="module")
def poseidon_setup(zk_ctx):
    cc = compile_circuit(circuits.PoseidonCircuit())
    pk, vk, td = groth16.setup(cc, 11, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, window_bits_g1=8, window_bits_g2=6)
    yield cc, pk, vk, td, prover
    prover.close()


def test_setup_matches_oracle(zk_ctx, poseidon_setup):
    """GPU fixed-base setup == C-oracle setup on the same trapdoor."""
    from oracle import cref
    cc, pk, vk, td, _ = poseidon_setup
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    pk2, vk2, td2 = groth16.setup(cc, 11, mul)
    assert td == td2
    for name in ("g1_a", "g1_b", "g1_k", "g1_z", "g2_b", "g1_alpha", "g2_delta"):
        assert np.array_equal(getattr(pk, name), getattr(pk2, name)), name
    assert np.array_equal(vk.g1_k, vk2.g1_k)


def test_poseidon_prove(zk_ctx, poseidon_setup):
    from oracle import cref, pyref
    cc, pk, vk, td, prover = poseidon_setup
    rng = random.Random(3)
    batch = 67
    datas = [297262668938251460872476410954775437897592223497, 0, 1, pyref.R - 1] + \
            [rng.randrange(pyref.R) for _ in range(batch - 4)]
    inputs = [cc.assignment_vector({"Data": d, "Hash": pyref.poseidon_hash([d])}) for d in datas]
    inp = np.stack([to_mont_array(v) for v in inputs])
    rs_int = [(rng.randrange(pyref.R), rng.randrange(pyref.R)) for _ in range(batch)]
    rs_int[1] = (0, 0)
    rs = np.stack([to_mont_array(v) for v in rs_int])
    # witness solve parity (wires and a, b, c)
    status, wires, abc = prover.solve(inp, want_wires=True, want_abc=True)
    assert not status.any()
    rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
    for i in (0, 1, 5, batch - 1):
        rc, w, a, b, c = cref.r1cs_solve(rh, inp[i])
        assert rc == 0
        assert np.array_equal(wires[i], w)
        assert np.array_equal(abc[0, i], a) and np.array_equal(abc[1, i], b)
        assert np.array_equal(abc[2, i], c)
    # prove parity, every proof
    proofs, status = prover.prove(inp, rs)
    assert not status.any()
    want, wstatus, _ = cref.groth16_prove_batch(rh, ph, inp, rs)
    assert not wstatus.any()
    assert np.array_equal(proofs, want)
    # closed form + pairing on two of them
    vkd = dict(alpha=_pts(np.concatenate([vk.g1_alpha, vk.g1_alpha, vk.g2_beta]))[0],
               beta=_pts(np.concatenate([vk.g1_alpha, vk.g1_alpha, vk.g2_beta]))[2],
               gamma=_pts(np.concatenate([vk.g1_alpha, vk.g1_alpha, vk.g2_gamma]))[2],
               delta=_pts(np.concatenate([vk.g1_alpha, vk.g1_alpha, vk.g2_delta]))[2],
               k=[_pts(np.concatenate([k, k, vk.g2_beta]))[0] for k in vk.g1_k])
    for i in (0, 1):
        w_int = from_mont_array(wires[i])
        exp = pyref.expected_proof(cc.constraints, cc.n_wires, cc.n_public, w_int, td, pk.log_n,
                                   *rs_int[i])
        assert _pts(proofs[i]) == exp
        assert pyref.verify(vkd, w_int[:cc.n_public], _pts(proofs[i]))


def test_invalid_witness_status(zk_ctx, poseidon_setup):
    cc, pk, vk, td, prover = poseidon_setup
    from oracle import pyref
    good = cc.assignment_vector({"Data": 5, "Hash": pyref.poseidon_hash([5])})
    bad = cc.assignment_vector({"Data": 5, "Hash": 7})
    inp = np.stack([to_mont_array(good), to_mont_array(bad), to_mont_array(good)])
    rs = np.stack([to_mont_array([1, 2])] * 3)
    proofs, status = prover.prove(inp, rs)
    assert list(status) == [0, -5, 0]
    assert np.array_equal(proofs[0], proofs[2])


@pytest.mark.parametrize("levels,populated,wbits", [(8, 3, (7, 5)), (24, 0, (7, 5)),
                                                    (8, 5, (109, 106)), (12, 2, (0, 0)),
                                                    (8, 4, (210, 207)), (8, 4, (311, 308))])
def test_smt_inclusion_prove(zk_ctx, levels, populated, wbits):
    from oracle import cref
    cc = compile_circuit(circuits.smt_inclusion_circuit(levels))
    pk, vk, td = groth16.setup(cc, 5, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, *wbits)
    rng = random.Random(levels)
    batch = 9
    ws = [smt_witness.synthetic_inclusion(rng, levels, populated) for _ in range(batch)]
    ws[2]["Root"] = (ws[2]["Root"] + 1) % H.R          # wrong root -> unsatisfied
    inp = np.stack([to_mont_array(cc.assignment_vector(w)) for w in ws])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in range(batch)])
    proofs, status = prover.prove(inp, rs)
    rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
    want, wstatus, _ = cref.groth16_prove_batch(rh, ph, inp, rs)
    assert list(status != 0) == list(wstatus != 0) == [i == 2 for i in range(batch)]
    ok = status == 0
    assert np.array_equal(proofs[ok], want[ok])
    prover.close()


def test_pipelined_submit_collect_matches_blocking(zk_ctx, poseidon_setup):
    """submit(k+1) before collect(k): same proofs as the blocking call, batches of different sizes,
    one unsatisfied witness, and the guard against mixing entry points while a batch is in flight."""
    from gnark_crypto_primitives_amd import lib
    from oracle import pyref
    cc, pk, vk, td, prover = poseidon_setup
    rng = random.Random(8)
    batches = []
    for bsz in (5, 70, 1, 64):
        datas = [rng.randrange(pyref.R) for _ in range(bsz)]
        inp = np.stack([to_mont_array(cc.assignment_vector(
            {"Data": d, "Hash": pyref.poseidon_hash([d])})) for d in datas])
        rs = np.stack([to_mont_array([rng.randrange(pyref.R), rng.randrange(pyref.R)])
                       for _ in range(bsz)])
        batches.append((inp, rs))
    batches[1][0][3, 0, 0] ^= 1
    want = [prover.prove(i, r) for i, r in batches]
    got = list(prover.prove_stream(batches))
    for (wp, ws), (gp, gs) in zip(want, got):
        assert np.array_equal(ws, gs)
        assert np.array_equal(wp[ws == 0], gp[gs == 0])
    assert list(got[1][1] != 0) == [i == 3 for i in range(70)]
    prover.submit(*batches[0])
    with pytest.raises(lib.ZkmiError):
        prover.solve(batches[0][0])
    prover.collect()


def _prove_and_check(zk_ctx, cc, assignments, seed, wbits=(7, 5), info_out=None, keys=None, **plan):
    """prove a batch on the GPU and compare every proof with the C oracle."""
    from oracle import cref
    pk, vk, td = keys if keys is not None else groth16.setup(cc, seed, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, *wbits, **plan)
    if info_out is not None:
        info_out.update(zk_ctx.pk_info(prover.pk_h))
    rng = random.Random(seed)
    inp = np.stack([to_mont_array(cc.assignment_vector(a)) for a in assignments])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in assignments])
    proofs, status = prover.prove(inp, rs)
    want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
    prover.close()
    assert np.array_equal(status != 0, wstatus != 0)
    ok = status == 0
    assert np.array_equal(proofs[ok], want[ok])
    return status


def test_config3_verifier_inclusion_and_exclusion(zk_ctx):
    """BASELINE config 3: smt.Verifier with fnc = 0 and fnc = 1 through one constraint system."""
    levels = 10
    cc = compile_circuit(circuits.smt_verifier_circuit(levels))
    rng = random.Random(33)
    asg = []
    for i in range(12):
        w = smt_witness.synthetic_inclusion(rng, levels, 3)
        if i % 2 == 0:
            asg.append(dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Fnc=0))
        else:
            other = (w["Key"] & 0b111) | (((w["Key"] >> 3) ^ 1) << 3)
            asg.append(dict(w, OldKey=w["Key"], OldValue=w["Value"], IsOld0=0, Key=other, Value=0,
                            Fnc=1))
    asg[5] = dict(asg[5], Key=asg[5]["OldKey"])      # excluded key is present -> unsatisfied
    status = _prove_and_check(zk_ctx, cc, asg, 3)
    assert list(status != 0) == [i == 5 for i in range(12)]


def test_config4_elgamal_add(zk_ctx):
    """BASELINE config 4: homomorphic add of two BabyJubJub ElGamal ciphertexts (domain 2^4)."""
    from gnark_crypto_primitives_amd.ecc import babyjub_native as bjj
    cc = compile_circuit(circuits.ElGamalAddCircuit())
    rng = random.Random(4)
    pub = bjj.mul(bjj.BASE, rng.randrange(bjj.ORDER))

    def enc(m):
        k = rng.randrange(bjj.ORDER)
        return bjj.mul(bjj.BASE, k) + bjj.add(bjj.mul(bjj.BASE, m), bjj.mul(pub, k))
    distinct = []
    for i in range(18):                       # 0.1 s each: Python curve arithmetic
        a, b = enc(i), enc(1000 - i)
        distinct.append({"A": list(a), "B": list(b),
                         "Sum": list(bjj.add(a[:2], b[:2]) + bjj.add(a[2:], b[2:]))})
    asg = [dict(distinct[i % 18]) for i in range(70)]
    asg[7]["Sum"] = list(asg[7]["Sum"])
    asg[7]["Sum"][0] = (asg[7]["Sum"][0] + 1) % H.R
    status = _prove_and_check(zk_ctx, cc, asg, 4, wbits=(6, 4))
    assert list(status != 0) == [i == 7 for i in range(70)]


def test_config4b_elgamal_encrypt(zk_ctx):
    """Encrypt circuit (elgamal/encrypt_test.go:61-86): two fixed-base and one variable-base
    scalar multiplication in-circuit, 7 232 constraints."""
    from gnark_crypto_primitives_amd.ecc import babyjub_native as bjj
    cc = compile_circuit(circuits.ElGamalEncryptCircuit())
    rng = random.Random(44)
    asg = []
    for k, m in ((12345, 67890), (1, 0), (rng.randrange(bjj.ORDER), rng.getrandbits(60))):
        pub = bjj.mul(bjj.BASE, rng.randrange(1, bjj.ORDER))
        ex = bjj.mul(bjj.BASE, k) + bjj.add(bjj.mul(bjj.BASE, m), bjj.mul(pub, k))
        asg.append({"PubKey": list(pub), "Expected": list(ex), "K": k, "M": m})
    status = _prove_and_check(zk_ctx, cc, asg, 5)
    assert not status.any()


def test_groth16_regression_fixture_gpu(zk_ctx):
    """GPU proofs equal the committed proof bytes of tests/golden/groth16_regression.json."""
    from tests.test_oracle import _regression_cases
    for name, circuit, fx in _regression_cases():
        cc = compile_circuit(circuit)
        assert cc.fingerprint() == fx["fingerprint"]
        pk, _, _ = groth16.setup(cc, fx["setup_seed"], groth16.gpu_mul(zk_ctx))
        prover = groth16.Prover(zk_ctx, cc, pk, 7, 5)
        inp = np.stack([to_mont_array([int(x) for x in v]) for v in fx["inputs"]])
        rs = np.stack([to_mont_array([int(x) for x in v]) for v in fx["rs"]])
        proofs, status = prover.prove(inp, rs)
        prover.close()
        assert not status.any()
        assert [p.tobytes().hex() for p in proofs] == fx["proofs_hex"]


def test_every_opcode_on_gpu(zk_ctx):
    """The circuit of tests/test_frontend.py::Mixed touches every witness-program opcode
    (ToBinary, IsZero, DivUnchecked, Inverse, Select, Lookup2, Xor, Or, ...): GPU solve and prove
    against the oracle, including unsatisfiable inputs (bad output, overflowing ToBinary, x/0)."""
    from tests.test_frontend import Mixed, _mixed_expected
    cc = compile_circuit(Mixed())
    rng = random.Random(77)
    asg = []
    for i in range(40):
        x = rng.randrange(1 << 16)
        y = x if i % 5 == 0 else rng.randrange(H.R)
        asg.append({"X": x, "Y": y, "Z": _mixed_expected(x, y)})
    asg[3]["Z"] = (asg[3]["Z"] + 1) % H.R
    asg[9] = {"X": 1 << 20, "Y": 7, "Z": 0}
    asg[11] = {"X": 5, "Y": H.R - 2, "Z": 0}
    status = _prove_and_check(zk_ctx, cc, asg, 12)
    assert list(status != 0) == [i in (3, 9, 11) for i in range(40)]


def test_config5_secp256k1_address(zk_ctx):
    """BASELINE config 5: ecdsa.DeriveAddress (one Keccak-f[1600] in R1CS: 151 945 constraints,
    domain 2^18) solved and proved on the GPU; every proof bit-exact against the C oracle, the
    address taken from the Python oracle (public vectors for keys 1 and 2)."""
    from gnark_crypto_primitives_amd.std.emulated import limbs_of
    from oracle import pyref
    cc = H.compiled("address")
    assert cc.domain_log2() == 18
    rng = random.Random(55)
    asg = []
    for priv in (1, 2, rng.randrange(1, pyref.SECP_N), rng.randrange(1, pyref.SECP_N)):
        pub = pyref.secp256k1_mul(priv)
        asg.append({"Address": pyref.eth_address(pub), "X": limbs_of(pub[0]),
                    "Y": limbs_of(pub[1])})
    assert asg[0]["Address"] == 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF
    asg[3] = dict(asg[3], Address=asg[2]["Address"])      # someone else's address -> unsatisfied
    keys = groth16.setup(cc, 5, groth16.gpu_mul(zk_ctx))      # one setup for both table plans
    status = _prove_and_check(zk_ctx, cc, asg, 5, wbits=(5, 4), keys=keys)
    assert list(status != 0) == [False, False, False, True]
    # the auto plan of a witness of bits (zkmi_pk_desc.sparse_witness = 2, what bench.py runs): small
    # subset-sum tables for the wire MSMs, the dense quotient MSM on its own sign-pattern tables
    assert cc.n_boolean_wires * 100 >= cc.n_wires * 99
    info = {}
    status = _prove_and_check(zk_ctx, cc, asg, 6, wbits=(0, 0), info_out=info, keys=keys,
                              max_batch=64)
    assert list(status != 0) == [False, False, False, True]
    assert info["g2_comb_k"] == 12 and info["g2_windows"] == 254, info      # wires: subset sums
    assert info["g1_comb_k"] >= 16 and info["g1_windows"] == 255, info      # quotient: sign patterns


def test_prove_empty_and_single(zk_ctx, poseidon_setup):
    """batch = 0 is a no-op; batch = 1 (63 padding lanes) equals the same proof inside a batch."""
    from oracle import pyref
    cc, pk, vk, td, prover = poseidon_setup
    n_in = cc.n_public - 1 + cc.n_secret
    proofs, status = prover.prove(np.zeros((0, n_in, 4), dtype=np.uint64),
                                  np.zeros((0, 2, 4), dtype=np.uint64))
    assert proofs.shape == (0, 32) and status.shape == (0,)
    inp = np.stack([to_mont_array(cc.assignment_vector({"Data": d, "Hash": pyref.poseidon_hash([d])}))
                    for d in (11, 12, 13)])
    rs = np.stack([to_mont_array([5 + i, 9 + i]) for i in range(3)])
    many, st = prover.prove(inp, rs)
    one, st1 = prover.prove(inp[1:2], rs[1:2])
    assert not st.any() and not st1.any()
    assert np.array_equal(one[0], many[1])


@pytest.mark.parametrize("wbits", [(7, 5), (105, 104), (0, 0), (204, 203), (304, 303)])
def test_degenerate_circuits(zk_ctx, wbits):
    """Smallest possible keys: one linear constraint between public inputs (no private wire at all:
    the K MSM is empty), and one product with a single internal wire (domain 2^1 / 2^0 edge)."""
    from gnark_crypto_primitives_amd.frontend import Public

    class Linear:
        X = Public()
        Y = Public()

        def define(self, api):
            api.AssertIsEqual(api.Add(self.X, 3), self.Y)

    class Square:
        X = Public()
        Y = Public()

        def define(self, api):
            api.AssertIsEqual(api.Mul(self.X, self.X), self.Y)

    cc = compile_circuit(Linear())
    assert cc.n_wires == 3
    status = _prove_and_check(zk_ctx, cc, [{"X": 4, "Y": 7}, {"X": 0, "Y": 3}, {"X": 1, "Y": 5}],
                              21, wbits)
    assert list(status != 0) == [False, False, True]
    cc = compile_circuit(Square())
    status = _prove_and_check(zk_ctx, cc, [{"X": 4, "Y": 16}, {"X": H.R - 1, "Y": 1},
                                           {"X": 2, "Y": 5}], 22, wbits)
    assert list(status != 0) == [False, False, True]


def test_prove_witness_batch_matches_oracle(zk_ctx):
    """zkmi_prove_witness_batch (the entry for a caller that keeps gnark's own solver): fed with
    the C oracle's solved (W, a, b, c) it yields the oracle's proofs -- and the proofs of
    zkmi_prove_batch on the same inputs.  The key is described the way gnark's ProvingKey is
    (InfinityA / InfinityB byte maps + nbPublic), under per-window, shared and comb plans."""
    from oracle import cref
    cc = compile_circuit(circuits.smt_inclusion_circuit(10))
    pk, vk, td = groth16.setup(cc, 41, groth16.gpu_mul(zk_ctx))
    rng = random.Random(41)
    batch = 70
    ws = [smt_witness.synthetic_inclusion(rng, 10, 4) for _ in range(batch)]
    inp = np.stack([to_mont_array(cc.assignment_vector(w)) for w in ws])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in range(batch)])
    rh, ph = cref.R1csHandle(cc), cref.PkHandle(pk)
    solved = [cref.r1cs_solve(rh, inp[i]) for i in range(batch)]
    assert all(s[0] == 0 for s in solved)
    W = np.stack([s[1] for s in solved])
    A, B, Cc = (np.stack([s[k] for s in solved]) for k in (2, 3, 4))
    assert W.shape == (batch, cc.n_wires, 4) and A.shape == (batch, cc.n_constraints, 4)
    want, wstatus, _ = cref.groth16_prove_batch(rh, ph, inp, rs)
    assert not wstatus.any()
    for wbits in ((7, 5), (108, 106), (0, 0), (209, 208), (310, 309)):
        prover = groth16.Prover(zk_ctx, cc, pk, *wbits, gnark_key_layout=True)
        got = prover.prove_witness(W, A, B, Cc, rs)
        assert np.array_equal(got, want), wbits
        full, status = prover.prove(inp, rs)
        assert not status.any() and np.array_equal(full, want)
        with pytest.raises(ValueError):
            prover.prove_witness(W[:, :-1], A, B, Cc, rs)       # wrong wire count
        prover.close()


def test_key_loaded_from_gnark_format_file(zk_ctx, tmp_path):
    """a proving key written in gnark's raw layout (gnark_io, [UPSTREAM-RECALL],
    parity unpinned) is read back, handed to zkmi_pk_load through gnark's own fields and proves the
    oracle's proofs; the proofs round-trip through Proof.WriteTo bytes and verify."""
    from gnark_crypto_primitives_amd import gnark_io, verify
    from oracle import cref, pyref
    cc = compile_circuit(circuits.PoseidonCircuit())
    pk, vk, _ = groth16.setup(cc, 43, groth16.gpu_mul(zk_ctx))
    gnark_io.save_key(str(tmp_path / "poseidon.pk"), pk, vk)
    pk2, vk2 = gnark_io.load_key(str(tmp_path / "poseidon.pk"), cc.n_public)
    prover = groth16.Prover(zk_ctx, cc, pk2, 8, 6, gnark_key_layout=True)
    datas = [3, 4, 5]
    hashes = [pyref.poseidon_hash([d]) for d in datas]
    inp = np.stack([to_mont_array(cc.assignment_vector({"Data": d, "Hash": h}))
                    for d, h in zip(datas, hashes)])
    rs = np.stack([to_mont_array([7 + i, 9 + i]) for i in range(3)])
    proofs, status = prover.prove(inp, rs)
    prover.close()
    want, _, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
    assert not status.any() and np.array_equal(proofs, want)
    back = gnark_io.proof_from_bytes(gnark_io.proof_to_bytes(proofs[1]))
    assert np.array_equal(back, proofs[1])
    assert verify.verify(vk2, [hashes[1]], back)


def test_pk_plan_respects_budget_and_batch(zk_ctx):
    """zkmi_pk_desc.table_budget_bytes caps the auto plan's tables; two keys live on one context;
    a key whose infinity maps disagree with its point counts is refused."""
    from gnark_crypto_primitives_amd import lib
    from oracle import cref
    cc = compile_circuit(circuits.smt_inclusion_circuit(24))
    pk, _, _ = groth16.setup(cc, 42, groth16.gpu_mul(zk_ctx))
    rng = random.Random(42)
    ws = [smt_witness.synthetic_inclusion(rng, 24, 5) for _ in range(5)]
    inp = np.stack([to_mont_array(cc.assignment_vector(w)) for w in ws])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in ws])
    want, _, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
    provers = []
    for budget in (1 << 28, 1 << 31):
        p = groth16.Prover(zk_ctx, cc, pk, 0, 0, table_budget_bytes=budget, max_batch=256,
                           msm_chunk_factor=4)
        info = zk_ctx.pk_info(p.pk_h)
        assert info["g1_table_bytes"] + info["g2_table_bytes"] <= budget, (budget, info)
        provers.append((p, info))
    assert provers[1][1]["g1_table_bytes"] > provers[0][1]["g1_table_bytes"]
    for p, _ in provers:                       # both keys resident at once
        proofs, status = p.prove(inp, rs)
        assert not status.any() and np.array_equal(proofs, want)
    for p, _ in provers:
        p.close()
    bad = groth16.ProvingKey()
    bad.__dict__.update(pk.__dict__)
    bad.a_wire = pk.a_wire[:-1]                # one retained wire fewer than points in g1_a
    with pytest.raises(lib.ZkmiError):
        class _P(groth16.Prover):
            pass
        k = groth16.Prover.__new__(groth16.Prover)
        inf_a, inf_b = bad.infinity_maps()
        keep = [np.ascontiguousarray(x) for x in (pk.g1_a, pk.g1_b, pk.g1_k, pk.g1_z, pk.g2_b,
                                                  pk.g1_alpha, pk.g1_beta, pk.g1_delta,
                                                  pk.g2_beta, pk.g2_delta)]
        pd = lib.PkDesc(pk.log_n, pk.n_wires, len(pk.a_wire), len(pk.b_wire), len(pk.k_wire),
                        pk.g1_z.shape[0], None, None, None, *[x.ctypes.data for x in keep], 7, 5,
                        inf_a.ctypes.data, inf_b.ctypes.data, cc.n_public, 0, 0, 0, 0)
        zk_ctx.pk_load(pd)
