"""Shared test helpers: seeded field/curve test data in gnark's memory layout."""
import random

import numpy as np

from gnark_crypto_primitives_amd.frontend.compile import (array_to_ints, ints_to_array,
                                                          to_mont_array)

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT_P = (1 << 256) % P
MONT_P_INV = pow(MONT_P, P - 2, P)

G1_GEN = (1, 2)
# BN254 G2 generator (x = x0 + x1*u, y = y0 + y1*u), the one gnark-crypto and EIP-197 use
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


def fq_mont(xs):
    return ints_to_array([x % P * MONT_P % P for x in xs])


def fq_unmont(a):
    return [v * MONT_P_INV % P for v in array_to_ints(a)]


def g1_gen_mont():
    return fq_mont(G1_GEN).reshape(-1)


def g2_gen_mont():
    return fq_mont([G2_GEN[0][0], G2_GEN[0][1], G2_GEN[1][0], G2_GEN[1][1]]).reshape(-1)


def rand_fr(rng, n, special=True):
    """n scalars < r, Montgomery; sprinkles edge values (0, 1, r-1, small, 2^k)."""
    xs = [rng.randrange(R) for _ in range(n)]
    if special and n >= 8:
        for i, v in enumerate([0, 1, R - 1, 2, (1 << 128), (1 << 253), R - 2, 12345]):
            xs[(i * 7919) % n] = v
    return xs, to_mont_array(xs)


def rng(seed):
    return random.Random(seed)


# ---- compiled circuits shared by the full-size GPU tests (compile once per session: Arbo-160 is 8 s,
# the address circuit 7 s, its SCS lowering 17 s)
import functools


@functools.lru_cache(maxsize=None)
def compiled(name):
    from gnark_crypto_primitives_amd import circuits
    from gnark_crypto_primitives_amd.frontend import compile_circuit
    if name == "arbo160":
        return compile_circuit(circuits.smt_inclusion_circuit(160))
    # 16 solver lanes is what the automatic choice arrives at for the Keccak circuits; naming it
    # skips the trial schedules at 4 and 8 lanes
    if name == "address":
        return compile_circuit(circuits.AddressCircuit(), 16)
    if name == "address-commit":
        return compile_circuit(circuits.AddressCircuitCommit(), 16)
    if name == "address-bytes":
        return compile_circuit(circuits.AddressCircuitByteTables(), 16)
    if name == "emulated-poseidon":
        return compile_circuit(circuits.EmulatedPoseidonCircuit(), 16)
    if name == "address-scs":
        from gnark_crypto_primitives_amd.frontend.scs import compile_scs
        return compile_scs(compiled("address"))
    raise KeyError(name)
