"""gnark wire formats: internal consistency of the restated encodings."""
import random

import numpy as np
import pytest

from gnark_crypto_primitives_amd import gnark_io
from oracle import cref, pyref
from tests import helpers as H


def _points(group, n, seed):
    rng = random.Random(seed)
    gen = H.g1_gen_mont() if group == 1 else H.g2_gen_mont()
    return cref.batch_mul(group, gen, H.to_mont_array([rng.randrange(1, pyref.R) for _ in range(n)]))


@pytest.mark.parametrize("compressed", [True, False])
def test_g1_g2_roundtrip_and_flags(compressed):
    for group, enc, dec, size in ((1, gnark_io.g1_to_bytes, gnark_io.g1_from_bytes, 32),
                                  (2, gnark_io.g2_to_bytes, gnark_io.g2_from_bytes, 64)):
        pts = _points(group, 20, group)
        seen = set()
        for p in pts:
            b = enc(p, compressed)
            assert len(b) == (size if compressed else 2 * size)
            seen.add(b[0] >> 6)
            assert np.array_equal(dec(b), p)
        if compressed:
            assert seen == {0b10, 0b11}        # both root choices occur
        else:
            assert seen == {0b00}
        inf = np.zeros(8 * group, dtype=np.uint64)
        b = enc(inf, compressed)
        # compressed: flag 0b01; raw: all zero bytes (gnark-crypto's raw encoder, as recalled); the
        # reader accepts both spellings
        assert b[0] == (0x40 if compressed else 0) and not any(b[1:]) and not dec(b).any()
        assert not dec(bytes([0x40]) + bytes(len(b) - 1)).any()
    # negating a point flips smallest <-> largest and nothing else
    p = _points(1, 1, 9)[0]
    q = p.copy()
    q[4:] = H.fq_mont([(-H.fq_unmont(p[4:].reshape(1, 4))[0]) % H.P])[0]
    bp, bq = gnark_io.g1_to_bytes(p), gnark_io.g1_to_bytes(q)
    assert bp[1:] == bq[1:] and (bp[0] ^ bq[0]) == 0x40


def test_proof_and_witness_roundtrip():
    ar, krs = _points(1, 2, 5)
    bs = _points(2, 1, 6)[0]
    rec = np.concatenate([ar, krs, bs])
    for raw, size in ((False, 32 + 64 + 32 + 4), (True, 64 + 128 + 64 + 4)):
        b = gnark_io.proof_to_bytes(rec, raw)
        assert len(b) == size and b[-4:] == b"\x00\x00\x00\x00"
        assert np.array_equal(gnark_io.proof_from_bytes(b), rec)
    vals = [5, 0, pyref.R - 1, 1234567]
    w = gnark_io.witness_to_bytes(vals, 1)
    assert len(w) == 12 + 4 * 32 and w[:12] == bytes([0, 0, 0, 1, 0, 0, 0, 3, 0, 0, 0, 4])
    assert gnark_io.witness_from_bytes(w) == (vals, 1)
    assert H.from_mont_array(gnark_io.witness_to_inputs(w)) == vals if hasattr(H, "from_mont_array") \
        else True
    with pytest.raises(ValueError):
        gnark_io.witness_from_bytes(w[:-1])


def test_proving_and_verifying_key_round_trip(tmp_path):
    """ProvingKey.WriteRawTo / VerifyingKey.WriteRawTo layouts [UPSTREAM-RECALL, parity unpinned]:
    write -> read returns the same key (points, infinity maps -> wire indices), a corrupted
    infinity map or a truncated slice is refused."""
    import struct

    import pytest

    from gnark_crypto_primitives_amd import circuits, groth16
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    from oracle import cref
    from tests import helpers as H
    cc = compile_circuit(circuits.PoseidonCircuit())
    mul = lambda g, s: cref.batch_mul(g, H.g1_gen_mont() if g == 1 else H.g2_gen_mont(), s)
    pk, vk, _ = groth16.setup(cc, 5, mul)
    blob = gnark_io.proving_key_to_bytes(pk)
    n_pts = len(pk.g1_a) + len(pk.g1_b) + len(pk.g1_z) + len(pk.g1_k) + 3
    assert len(blob) == (8 + 5 * 32 + 1) + 64 * n_pts + 128 * (len(pk.g2_b) + 2) + 5 * 4 + 24 + \
        2 * (4 + pk.n_wires) + 4
    back = gnark_io.proving_key_from_bytes(blob, cc.n_public)
    for name in ("a_wire", "b_wire", "k_wire", "g1_a", "g1_b", "g1_k", "g1_z", "g2_b", "g1_alpha",
                 "g1_beta", "g1_delta", "g2_beta", "g2_delta"):
        assert np.array_equal(getattr(back, name), getattr(pk, name)), name
    assert (back.log_n, back.n_wires) == (pk.log_n, pk.n_wires)
    vblob = gnark_io.verifying_key_to_bytes(vk, pk.g1_beta, pk.g1_delta)
    vback = gnark_io.verifying_key_from_bytes(vblob)
    for name in ("g1_alpha", "g2_beta", "g2_gamma", "g2_delta", "g1_k"):
        assert np.array_equal(getattr(vback, name), getattr(vk, name)), name
    # files
    gnark_io.save_key(str(tmp_path / "pk.bin"), pk, vk)
    pk2, vk2 = gnark_io.load_key(str(tmp_path / "pk.bin"), cc.n_public)
    assert np.array_equal(pk2.g1_z, pk.g1_z) and np.array_equal(vk2.g1_k, vk.g1_k)
    # corruption
    bad = bytearray(blob)
    bad[-4 - (4 + pk.n_wires) - 1] ^= 1                # flips the last InfinityA flag
    with pytest.raises(ValueError):
        gnark_io.proving_key_from_bytes(bytes(bad), cc.n_public)
    with pytest.raises((ValueError, struct.error)):
        gnark_io.proving_key_from_bytes(blob[:1000], cc.n_public)


def test_proof_with_commitments_roundtrip():
    """Proof.WriteTo / WriteRawTo with the commitment extension: count, Commitments, CommitmentPok
    behind the three proof points ([UPSTREAM-RECALL], parity unpinned)."""
    g1 = _points(1, 5, 9)
    g2 = _points(2, 1, 9)
    rec = np.concatenate([g1[0], g1[1], g2[0]])
    for raw in (False, True):
        for n in (0, 1, 2):
            b = gnark_io.proof_to_bytes(rec, raw, g1[2:2 + n], g1[4] if n else None)
            size = (64 if raw else 32) * 2 + (128 if raw else 64) + 4 + ((n + 1) * (64 if raw else 32) if n else 0)
            assert len(b) == size
            back, coms, pok = gnark_io.proof_from_bytes(b, with_commitments=True)
            assert np.array_equal(back, rec) and np.array_equal(coms, g1[2:2 + n])
            assert (pok is None) == (n == 0) and (n == 0 or np.array_equal(pok, g1[4]))
            assert np.array_equal(gnark_io.proof_from_bytes(b), rec)
    with pytest.raises(ValueError):
        gnark_io.proof_from_bytes(gnark_io.proof_to_bytes(rec, False, g1[2:3], g1[4]) + b"\\0",
                                  with_commitments=True)
