"""PLONK arithmetisation ("sparse constraint system") of a compiled circuit.

gnark compiles a circuit for PLONK with ``frontend.Compile(field, scs.NewBuilder, circuit)``
(third-party module, go.mod:8) [UPSTREAM-RECALL]: every constraint is one gate
    qL.a + qR.b + qO.c + qM.a.b + qC (+ PI) = 0
over three wire columns, plus copy constraints tying equal wires together.  BASELINE config 5 names
that backend; the reference's own test of the circuit uses Groth16 over R1CS
(ecc/secp256k1/ecdsa/address_test.go:40,57), so nothing in the reference pins a PLONK result:
parity unpinned.

This module lowers the SSA witness program the R1CS frontend records (frontend/api.py: one field
operation per API call, dead code eliminated) to gates, one per operation, the SSA values being the
PLONK wires:

    d = a + b      ->  (a, b, d)  qL = 1, qR = 1, qO = -1        d = a * b   ->  qM = 1, qO = -1
    d = a - b      ->  qL = 1, qR = -1, qO = -1                 d = k a     ->  (a, a, d) qL = k, qO = -1
    d = a + k      ->  (a, a, d) qL = 1, qC = k, qO = -1         d = -a      ->  qL = 1, qO = 1
    d = a xor b    ->  qL = 1, qR = 1, qM = -2, qO = -1          d = k       ->  (d, d, d) qO = -1, qC = k
    R1CS row (L, R, O) kept as a row (assertions, hint-defining rows): (vL, vR, vO) qM = 1, qO = -1
    public input x_j: (v_j, v_j, v_j) qL = 1, PI_j = -x_j;  ONE wire: qL = 1, qC = -1

Hints (inverse-or-zero, bit decomposition, unchecked division, the batched inversion) get no gate:
as in the R1CS, the rows that use their outputs constrain them.  The witness program of the system
is the same program with an OP_ABC row per gate, so the GPU solver (csrc/solve.hip) emits the
three wire columns a, b, c directly.
"""
from __future__ import annotations

import numpy as np

from .api import (OP_ABC, OP_ADD, OP_ADDC, OP_BATCHINV, OP_BITS, OP_COPY, OP_DIV, OP_INV, OP_MUL,
                  OP_MULABC, OP_MULC, OP_NEG, OP_PAIR, OP_SETC, OP_SUB, OP_XOR, OP_XORABC, R)
from .compile import CompiledCircuit, build_vprogram

COSET_SHIFT = 5          # fr.MultiplicativeGen: the columns' identity permutation is X, 5X, 25X


class ScsCircuit:
    """Selectors, wiring and witness program of one circuit (gnark: constraint.SparseR1CS)."""

    def __init__(self, cc: CompiledCircuit, lanes_per_proof: int = 0):
        self.cc = cc
        self.layout = cc.layout
        self.n_public = cc.n_public            # including ONE, as in the R1CS
        self.n_secret = cc.n_secret
        self.n_inputs = cc.n_inputs
        self.n_wires = cc.n_wires              # value slots below n_wires stay wire-backed
        self.consts = list(cc.consts)
        ops, val_wire = cc._ops, cc._val_wire
        n_pub = cc.n_public - 1
        gates = []                             # (a, b, c, qL, qR, qO, qM, qC) on SSA value ids
        one = 0                                # SSA value of the ONE wire (api.val_one)
        for j in range(1, n_pub + 1):          # public inputs first (their values are vals 1..)
            gates.append((j, j, j, 1, 0, 0, 0, 0))
        gates.append((one, one, one, 1, 0, 0, 0, R - 1))
        prog, row_of, chk = [], {}, {}

        def gate(g, check=0):
            row_of[len(prog)] = len(gates)
            chk[len(prog)] = check
            prog.append((OP_ABC, g[0], g[1], g[2]))
            gates.append(g)

        C = self.consts
        k_row = 0
        for op, d, a, b in ops:
            if op == OP_ABC:
                gate((d, a, b, 0, 0, R - 1, 1, 0), 1 if cc._row_check[k_row] else 0)
                k_row += 1
                continue
            if op == OP_MULABC:
                k_row += 1
                prog.append((OP_MUL, d, a, b))
                gate((a, b, d, 0, 0, R - 1, 1, 0))
            elif op == OP_XORABC:
                k_row += 1
                prog.append((OP_XOR, d, a, b))
                gate((a, b, d, 1, 1, R - 1, R - 2, 0))
            elif op == OP_MUL:
                prog.append((op, d, a, b))
                gate((a, b, d, 0, 0, R - 1, 1, 0))
            elif op == OP_ADD:
                prog.append((op, d, a, b))
                gate((a, b, d, 1, 1, R - 1, 0, 0))
            elif op == OP_SUB:
                prog.append((op, d, a, b))
                gate((a, b, d, 1, R - 1, R - 1, 0, 0))
            elif op == OP_MULC:
                prog.append((op, d, a, b))
                gate((a, a, d, C[b], 0, R - 1, 0, 0))
            elif op == OP_ADDC:
                prog.append((op, d, a, b))
                gate((a, a, d, 1, 0, R - 1, 0, C[b]))
            elif op == OP_NEG:
                prog.append((op, d, a, b))
                gate((a, a, d, 1, 0, 1, 0, 0))
            elif op == OP_COPY:
                prog.append((op, d, a, b))
                gate((a, a, d, 1, 0, R - 1, 0, 0))
            elif op == OP_SETC:
                prog.append((op, d, a, b))
                gate((d, d, d, 0, 0, R - 1, 0, C[b]))
            elif op in (OP_INV, OP_DIV, OP_BITS, OP_BATCHINV, OP_PAIR):
                prog.append((op, d, a, b))     # hints: constrained by the rows that use them
            else:
                raise AssertionError(op)
        # rows of the public-input / ONE gates come first: prepend their OP_ABC rows
        head = [(OP_ABC, g[0], g[1], g[2]) for g in gates[:n_pub + 1]]
        shift = len(head)
        row_of = {i + shift: r for i, r in row_of.items()}
        chk = {i + shift: c for i, c in chk.items()}
        for i in range(shift):
            row_of[i] = i
        prog = head + prog
        self.n_gates = len(gates)
        self.n_constraints = self.n_gates      # rows the solver emits
        self.log_n = max(2, (self.n_gates - 1).bit_length())
        self.gates = gates
        (self.vprogram, self.v_n_rows, self.v_n_steps, self.v_n_slots, self.lanes_per_proof,
         self.schedule_cost) = build_vprogram(prog, val_wire, row_of, chk, cc.n_wires,
                                              lanes_per_proof)
        self._build_tables()

    # the CPU evaluation of the scheduled program is the R1CS one (same machinery)
    run_vprogram = CompiledCircuit.run_vprogram

    def assignment_vector(self, assignment):
        return self.cc.assignment_vector(assignment)

    def _build_tables(self):
        n = 1 << self.log_n
        g = self.gates
        col = lambda j: np.array([x[j] for x in g] + [0] * (n - len(g)), dtype=object)
        self.qL, self.qR, self.qO, self.qM, self.qC = (col(j) for j in (3, 4, 5, 6, 7))
        # copy constraints: positions (column, row) holding the same SSA value form one cycle
        wires = [[x[j] for x in g] for j in (0, 1, 2)]
        sigma = np.arange(3 * n, dtype=np.int64)          # identity on padding rows
        where = {}
        for c in range(3):
            for r, v in enumerate(wires[c]):
                where.setdefault(v, []).append(c * n + r)
        for pos in where.values():
            for i, p in enumerate(pos):
                sigma[p] = pos[(i + 1) % len(pos)]
        self.sigma = sigma
        self.wires = wires
        # Rows of each column whose value is known to be a bit (a wire the builder knows boolean,
        # the ONE wire, padding rows): the Lagrange-basis commitment of a column (plonk.py) orders
        # its bases with the other rows first, so that whole groups of its subset-sum tables see
        # one-bit scalars.  Only speed depends on this classification, never a result.
        val_wire = self.cc._val_wire
        bw = self.cc.boolean_wires
        bitv = {0} | {v for v, w in val_wire.items() if w in bw}
        C = self.consts
        for op, d, a, b in self.cc._ops:          # bits stay bits under and / xor / not / copy
            if op in (OP_MUL, OP_MULABC, OP_XOR, OP_XORABC):
                if a in bitv and b in bitv:
                    bitv.add(d)
            elif op == OP_SUB:
                if a == 0 and b in bitv:          # 1 - b
                    bitv.add(d)
            elif op == OP_COPY:
                if a in bitv:
                    bitv.add(d)
            elif op == OP_SETC:
                if C[b] in (0, 1):
                    bitv.add(d)
        is_bit = lambda v: v in bitv
        self.lag_order = []
        for c in range(3):
            head = [r for r, v in enumerate(wires[c]) if not is_bit(v)]
            hs = set(head)
            self.lag_order.append(np.array(head + [r for r in range(n) if r not in hs],
                                           dtype=np.uint32))
            self.n_nonbit_rows = max(getattr(self, "n_nonbit_rows", 0), len(head))

    def is_satisfied(self, a, b, c, public):
        """gate equations + copy constraints on full columns (lists of ints, length n_gates)."""
        g = self.gates
        n_pub = self.n_public - 1
        for i, (wa, wb, wc, qL, qR, qO, qM, qC) in enumerate(g):
            pi = (-public[i]) % R if i < n_pub else 0
            if (qL * a[i] + qR * b[i] + qO * c[i] + qM * a[i] * b[i] + qC + pi) % R:
                return False, i
        n = 1 << self.log_n
        cols = (a, b, c)
        val = lambda p: cols[p // n][p % n] if p % n < len(g) else 0
        for p in range(3 * n):
            if self.sigma[p] != p and val(p) != val(int(self.sigma[p])):
                return False, -p
        return True, -1


def compile_scs(circuit, lanes_per_proof: int = 0) -> ScsCircuit:
    """``frontend.Compile(field, scs.NewBuilder, circuit)`` look-alike."""
    from .compile import compile_circuit
    cc = circuit if isinstance(circuit, CompiledCircuit) else compile_circuit(circuit)
    if getattr(cc, "commitments", None):
        # gnark's PLONK carries api.Commit through an extra selector column (Qcp) and commitments to
        # the committed wires' Lagrange polynomials [UPSTREAM-RECALL]: not built for this lowering
        raise NotImplementedError("circuits that call api.Commit are Groth16 / R1CS only here")
    return ScsCircuit(cc, lanes_per_proof)
