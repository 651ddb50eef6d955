"""Off-circuit Poseidon-BN254 (iden3 compatible), optimised form, Python ints.

The reference computes expected values off-circuit with iden3's poseidon.Hash
(hash/native/bn254/poseidon/poseidon_test.go:41) and carries a native restatement of the same
permutation in hash/emulated/bn254/poseidon/poseidon_test.go:112-273; this is that function for
witness generation (tree.smt_witness, bench.py).  Same tables and round structure as the gadget
(poseidon.py / poseidon.go:116-183).
"""
from .poseidon_constants import N_ROUNDS_F, N_ROUNDS_P, R, opt_params


def hash(inputs):
    t = len(inputs) + 1
    if not 2 <= t <= 17:
        raise ValueError("poseidon: 1..16 inputs")
    rp, rf2 = N_ROUNDS_P[t - 2], N_ROUNDS_F // 2
    c, m, p, s = opt_params(t)
    st = [0] + [int(x) % R for x in inputs]
    st = [(x + c[i]) % R for i, x in enumerate(st)]

    def mix(v, mat):
        return [sum(mat[j][i] * v[j] for j in range(t)) % R for i in range(t)]
    for r in range(rf2 - 1):
        st = [(pow(x, 5, R) + c[(r + 1) * t + i]) % R for i, x in enumerate(st)]
        st = mix(st, m)
    st = [(pow(x, 5, R) + c[rf2 * t + i]) % R for i, x in enumerate(st)]
    st = mix(st, p)
    for r in range(rp):
        s0 = (pow(st[0], 5, R) + c[(rf2 + 1) * t + r]) % R
        base = (2 * t - 1) * r
        new0 = (s[base] * s0 + sum(s[base + j] * st[j] for j in range(1, t))) % R
        st = [new0] + [(st[k] + s0 * s[base + t + k - 1]) % R for k in range(1, t)]
    for r in range(rf2 - 1):
        st = [(pow(x, 5, R) + c[(rf2 + 1) * t + rp + r * t + i]) % R for i, x in enumerate(st)]
        st = mix(st, m)
    st = [pow(x, 5, R) for x in st]
    return sum(m[j][0] * st[j] for j in range(t)) % R


def multihash(inputs):
    """poseidon.MultiHash semantics off-circuit (poseidon.go:54-91)."""
    inputs = list(inputs)
    if len(inputs) <= 16:
        return hash(inputs)
    return multihash([hash(inputs[i:i + 16]) for i in range(0, len(inputs), 16)])
