#!/usr/bin/env python3
"""One-off GPU check of tree/smt_emulated.py on the real emulated field (too slow for the suites):
the two-level inclusion verifier (four emulated Poseidon hashes), five proofs -- one with a wrong
value --, proof + commitment + proof of knowledge against the C oracle.
usage (GPU box): python3 tools/emulated_smt_gpu_check.py > gpurun_out/emulated_smt.log"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (before the library: one HIP runtime per process)

from gnark_crypto_primitives_amd import lib  # noqa: E402
from gnark_crypto_primitives_amd.frontend import compile_circuit  # noqa: E402
from gnark_crypto_primitives_amd.std import emulated as em  # noqa: E402
from gnark_crypto_primitives_amd.tree import smt_witness  # noqa: E402
from tests.test_gpu_commitment import _check  # noqa: E402
from tests.test_smt_emulated import EmulatedInclusion  # noqa: E402

t0 = time.time()
cc = compile_circuit(EmulatedInclusion(), 16)
print(f"compiled: {cc.n_constraints} constraints, {cc.n_wires} wires, domain 2^{cc.domain_log2()}, "
      f"{cc.v_n_steps} steps, {time.time() - t0:.0f} s", flush=True)
rng = random.Random(3)
v = lambda n: em.ValueOf(n, em.BN254Fr)
asg = []
for i in range(5):
    w = smt_witness.synthetic_inclusion(rng, 2, 1)
    asg.append({"Root": v(w["Root"]), "Key": v(w["Key"]), "Value": v(w["Value"] ^ (1 if i == 3 else 0)),
                "S0": v(w["Siblings"][0]), "S1": v(w["Siblings"][1])})
ctx = lib.Context(0)
t0 = time.time()
_check(ctx, cc, asg, [3], 91, wbits=(0, 0), publics=[list(a["Root"]) for a in asg], max_batch=64)
print(f"GPU == C oracle on proofs, commitments, proofs of knowledge; lane 3 unsatisfied on both; "
      f"the verifier accepts ({time.time() - t0:.0f} s)")
ctx.close()
