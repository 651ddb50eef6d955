"""Off-circuit assignment builder for the SMT verifier circuits.

The reference builds these records from a live Arbo tree (tree/smt/wrapper_arbo.go:13-184,
testutil/utils.go:95-185): key and value as little-endian integers, siblings unpacked root->leaf
and zero-padded to the circuit depth.  Arbo and PebbleDB are not part of the prover hot path; this
module produces records of exactly that shape from a synthetic path:
k populated siblings at levels 0..k-1, the rest zero, root folded bottom-up with the circuit's own
rule (tree/smt/verifier_level.go:8-17, tree/smt/utils.go:50-55; bit i of the key steers level i).
"""
from ..hash import poseidon_native

R = poseidon_native.R


def root_from_path(key, value, siblings, hasher=poseidon_native.hash):
    n = len(siblings)
    lev = max([i for i in range(n) if siblings[i] % R], default=-1) + 1
    cur = hasher([key, value, 1])
    for i in range(lev - 1, -1, -1):
        cur = hasher([siblings[i], cur]) if (key >> i) & 1 else hasher([cur, siblings[i]])
    return cur


def synthetic_inclusion(rng, levels, populated, key_bits=None, value_bits=64):
    """One assignment of the shape of tree/test/verifier_bn254_test.go:23-34:
    {Root, Key, Value, Siblings[levels]}."""
    if not 0 <= populated < levels:
        raise ValueError("the last sibling must be zero (tree/smt/lev_ins.go:71)")
    key = rng.getrandbits(key_bits or levels)
    value = rng.getrandbits(value_bits)
    sib = [rng.randrange(1, R) for _ in range(populated)] + [0] * (levels - populated)
    return {"Root": root_from_path(key, value, sib), "Key": key, "Value": value, "Siblings": sib}


def synthetic_exclusion_empty(rng, levels, populated, hasher=poseidon_native.hash):
    """Exclusion proof against an EMPTY branch (tree/smt/verifier.go:66-81 with isOld0 = 1): the
    path of ``Key`` ends in the zero leaf after ``populated`` non-zero siblings, the root is the
    fold of that zero leaf.  Shape of an smt.Verifier assignment."""
    if not 1 <= populated < levels:
        raise ValueError("need 1 <= populated < levels")
    key = rng.getrandbits(levels)
    sib = [rng.randrange(1, R) for _ in range(populated)] + [0] * (levels - populated)
    cur = 0
    for i in range(populated - 1, -1, -1):
        cur = hasher([sib[i], cur]) if (key >> i) & 1 else hasher([cur, sib[i]])
    return dict(Root=cur, OldKey=0, OldValue=0, IsOld0=1, Key=key, Value=0, Fnc=1, Siblings=sib)
