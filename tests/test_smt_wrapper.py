"""smt.WrapperArbo restated over an in-memory Arbo-shaped tree (tree/smt/wrapper_arbo.go): every
assignment it emits must satisfy the circuits it is meant for -- Proof -> smt.Verifier (inclusion
and exclusion), Set/SetProof -> smt.Processor (insert next to an empty node, insert next to an
existing leaf = the "drop the last sibling" rule of wrapper_arbo.go:170-172, update), plus the
circomlib delete that Arbo lacks."""
import random

import pytest

from gnark_crypto_primitives_amd import circuits
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.hash import poseidon_native
from gnark_crypto_primitives_amd.tree import smt_witness
from gnark_crypto_primitives_amd.tree.smt_wrapper import (HASH_LEN, KeyNotFound, MemTree,
                                                          WrapperArbo, big_int_to_bytes,
                                                          bytes_to_big_int, delete_assignment)

LEVELS = 10


def processor_inputs(a):
    return dict(OldRoot=a.OldRoot, Siblings=a.Siblings, OldKey=a.OldKey, OldValue=a.OldValue,
                IsOld0=a.IsOld0, NewKey=a.NewKey, NewValue=a.NewValue, Fnc0=a.Fnc0, Fnc1=a.Fnc1,
                NewRoot=a.NewRoot)


def verifier_inputs(a):
    return dict(Root=a.OldRoot, OldKey=a.OldKey, OldValue=a.OldValue, IsOld0=a.IsOld0,
                Key=a.NewKey, Value=a.NewValue, Fnc=a.Fnc0, Siblings=a.Siblings)


def test_arbo_byte_conventions():
    assert big_int_to_bytes(4, 0x0102) == b"\x02\x01\x00\x00"
    assert bytes_to_big_int(b"\x02\x01\x00\x00") == 0x0102 and bytes_to_big_int(b"") == 0
    with pytest.raises(ValueError):
        big_int_to_bytes(1, 256)
    t = MemTree(8)
    with pytest.raises(ValueError):
        t.Add(big_int_to_bytes(HASH_LEN, 256), big_int_to_bytes(HASH_LEN, 1))   # 9-bit key
    # single leaf: the root is the leaf hash; Get of another key returns that leaf
    t.Add(big_int_to_bytes(HASH_LEN, 5), big_int_to_bytes(HASH_LEN, 9))
    assert bytes_to_big_int(t.Root()) == poseidon_native.hash([5, 9, 1])
    with pytest.raises(KeyNotFound) as e:
        t.Get(big_int_to_bytes(HASH_LEN, 4))
    assert bytes_to_big_int(e.value.leaf[0]) == 5
    # keys 5 (101b) and 1 (001b) share bits 0, 1: two empty siblings, then each other
    t.Add(big_int_to_bytes(HASH_LEN, 1), big_int_to_bytes(HASH_LEN, 7))
    _, _, sibs, exists = t.GenProof(big_int_to_bytes(HASH_LEN, 1))
    assert exists and [bytes_to_big_int(s) for s in sibs] == [0, 0, poseidon_native.hash([5, 9, 1])]
    assert bytes_to_big_int(t.Root()) == smt_witness.root_from_path(
        1, 7, [0, 0, poseidon_native.hash([5, 9, 1])])


def test_wrapper_assignments_satisfy_the_circuits():
    from tests.test_elgamal_processor import _processor_circuit
    proc = compile_circuit(_processor_circuit(LEVELS))
    ver = compile_circuit(circuits.smt_verifier_circuit(LEVELS))
    rng = random.Random(10)
    w = WrapperArbo(MemTree(LEVELS), LEVELS)
    kinds = set()
    keys = {}
    inserts = []
    for step in range(40):
        if keys and step % 4 == 3:
            k = rng.choice(sorted(keys))           # update
        else:
            k = rng.randrange(1 << LEVELS)
        v = rng.randrange(1, 1 << 64)
        dry = w.SetProof(k, v)
        root_before = w.tree.root
        assert w.tree.root == root_before          # SetProof discards
        a = w.Set(k, v)
        assert a == dry
        assert (a.Fnc0, a.Fnc1) == ((0, 1) if k in keys else (1, 0))
        kinds.add((a.Fnc0, a.Fnc1, a.IsOld0))
        keys[k] = v
        proc.run_program(proc.assignment_vector(processor_inputs(a)))
        assert proc.last_status == 0, (step, a)
        if (a.Fnc0, a.Fnc1) == (1, 0):
            inserts.append(a)
        # tampering with the new root must fail
        if step % 7 == 0:
            bad = processor_inputs(a)
            bad["NewRoot"] = (a.NewRoot + 1) % poseidon_native.R
            proc.run_program(proc.assignment_vector(bad))
            assert proc.last_status != 0
    # insert into an empty slot, insert next to an existing leaf (last sibling dropped), update
    assert kinds >= {(1, 0, 1), (1, 0, 0), (0, 1, 0)}
    # circomlib delete = the inverse of an insert
    for a in inserts[:6]:
        d = delete_assignment(a)
        proc.run_program(proc.assignment_vector(processor_inputs(d)))
        assert proc.last_status == 0
    # Proof: membership of present keys, non-membership of absent ones (both isOld0 flavours)
    seen = set()
    for k in list(keys)[:8] + [rng.randrange(1 << LEVELS) for _ in range(30)]:
        a = w.Proof(k)
        assert a.Fnc0 == (0 if k in keys else 1)
        if k in keys:
            assert a.NewValue == keys[k] and a.OldKey == k
        seen.add((a.Fnc0, a.IsOld0))
        ver.run_program(ver.assignment_vector(verifier_inputs(a)))
        assert ver.last_status == 0, a
    assert (0, 0) in seen and (1, 0) in seen
    # a fresh tree: exclusion against an empty path (IsOld0 = 1)
    a = WrapperArbo(MemTree(LEVELS), LEVELS).Proof(77)
    assert (a.Fnc0, a.IsOld0, a.OldRoot) == (1, 1, 0)
