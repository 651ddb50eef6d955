"""``test.Assert`` of gnark (github.com/consensys/gnark/test), the harness every test of the
reference is written against: ``assert.SolvingSucceeded`` (tree/test/verifier_bn254_test.go:67,
hash/native/bn254/poseidon/poseidon_test.go:43,90), ``assert.ProverSucceeded``
(hash/emulated/bn254/poseidon/poseidon_test.go:72,103) and ``assert.CheckCircuit(circuit,
test.WithValidAssignment(..), test.WithInvalidAssignment(..))`` (elgamal/ciphertext_test.go:338,
tree/smt/processor_test.go:64, tree/smt/utils_test.go:33,41) [UPSTREAM-RECALL for gnark's side].

Here the constraint solver is the GPU witness solver (zkmi_solve_batch), the prover is the GPU
Groth16 prover (zkmi_prove_batch) and proofs are checked with the host verifier (verify.py), i.e.
what gnark does under its ``prover_checks`` build tag.  No CPU fallback: a ``lib.Context`` (a GPU)
is required.
"""
from __future__ import annotations

import random

import numpy as np

from . import groth16, lib, verify
from .frontend import compile_circuit
from .frontend.compile import CompiledCircuit, from_mont_array, to_mont_array

R = verify.R


class AssertionFailed(AssertionError):
    pass


class Assert:
    """One per test (gnark: ``assert := test.NewAssert(t)``).  Compiled circuits, keys and provers
    are cached per circuit object for the life of the Assert; ``close()`` releases the GPU side."""

    def __init__(self, ctx: lib.Context, setup_seed: int = 1, prover_seed: int = 2,
                 window_bits=(7, 5)):
        self.ctx = ctx
        self.setup_seed = setup_seed
        self._rng = random.Random(prover_seed)
        self._window_bits = window_bits
        self._cache = {}

    # ------------------------------------------------------------------ plumbing
    def _entry(self, circuit):
        key = id(circuit) if not isinstance(circuit, CompiledCircuit) else id(circuit)
        e = self._cache.get(key)
        if e is None:
            cc = circuit if isinstance(circuit, CompiledCircuit) else compile_circuit(circuit)
            pk, vk, _ = groth16.setup(cc, self.setup_seed, groth16.gpu_mul(self.ctx))
            prover = groth16.Prover(self.ctx, cc, pk, *self._window_bits)
            e = self._cache[key] = (cc, pk, vk, prover, circuit)
        return e

    def close(self):
        for _, _, _, prover, _ in self._cache.values():
            prover.close()
        self._cache.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _inputs(self, cc, assignments):
        return np.stack([to_mont_array(cc.assignment_vector(a)) for a in assignments])

    # ------------------------------------------------------------------ gnark's assertions
    def solve(self, circuit, assignments):
        """-> (status per assignment, wires) from the GPU solver."""
        cc, _, _, prover, _ = self._entry(circuit)
        status, wires, _ = prover.solve(self._inputs(cc, assignments), want_wires=True)
        return status, wires

    def SolvingSucceeded(self, circuit, *assignments):
        """gnark: compiles, runs the solver on the full witness, fails the test on any error.  The
        solved wires are also checked against every constraint on the host."""
        cc = self._entry(circuit)[0]
        status, wires = self.solve(circuit, assignments)
        for i, a in enumerate(assignments):
            if status[i] != 0:
                raise AssertionFailed(f"solving failed for assignment {i} (status {status[i]})")
            ok, row = cc.is_satisfied(from_mont_array(wires[i]))
            if not ok:
                raise AssertionFailed(f"assignment {i}: solver output violates constraint {row}")

    def SolvingFailed(self, circuit, *assignments):
        status, _ = self.solve(circuit, assignments)
        for i in range(len(assignments)):
            if status[i] == 0:
                raise AssertionFailed(f"solving succeeded for invalid assignment {i}")

    def prove(self, circuit, assignments):
        cc, _, vk, prover, _ = self._entry(circuit)
        inp = self._inputs(cc, assignments)
        rs = np.stack([to_mont_array([self._rng.randrange(R), self._rng.randrange(R)])
                       for _ in assignments])
        proofs, status = prover.prove(inp, rs)
        return proofs, status, inp

    def ProverSucceeded(self, circuit, *assignments):
        """Setup (cached), GPU prove, host verify against the assignment's public inputs."""
        cc, _, vk, _, _ = self._entry(circuit)
        proofs, status, inp = self.prove(circuit, assignments)
        n_pub = cc.n_public - 1
        for i in range(len(assignments)):
            if status[i] != 0:
                raise AssertionFailed(f"prover rejected assignment {i} (status {status[i]})")
            public = from_mont_array(inp[i, :n_pub])
            if not verify.verify(vk, public, proofs[i]):
                raise AssertionFailed(f"proof {i} does not verify")

    def ProverFailed(self, circuit, *assignments):
        _, status, _ = self.prove(circuit, assignments)
        for i in range(len(assignments)):
            if status[i] == 0:
                raise AssertionFailed(f"prover accepted invalid assignment {i}")

    def CheckCircuit(self, circuit, valid=(), invalid=(), prove=True):
        """test.Assert.CheckCircuit with WithValidAssignment / WithInvalidAssignment: every valid
        assignment must solve (and, with ``prove``, yield a verifying proof), every invalid one
        must be rejected by solver and prover."""
        if valid:
            self.SolvingSucceeded(circuit, *valid)
            if prove:
                self.ProverSucceeded(circuit, *valid)
        if invalid:
            self.SolvingFailed(circuit, *invalid)
            if prove:
                self.ProverFailed(circuit, *invalid)
