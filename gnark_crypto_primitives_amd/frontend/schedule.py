"""Static VLIW schedule of a witness program: S lanes of one wavefront work on ONE proof.

The witness program of a gadget circuit is a long dependent chain (Arbo-160: 162 Poseidon
permutations back to back, ~53 k dependent field operations out of 157 k), and one lane per proof
leaves the wavefront that runs it alone on its SIMD: the solve is the latency of that chain.  The
scheduler below packs independent operations of the same class into *steps* of up to S operations;
csrc/solve.hip runs a step with S sub-lanes per proof (64 / S proofs per wavefront), every sub-lane
executing its own (dst, a, b) of the step.  gnark's solver does the analogous thing on the CPU: it
levelises the instructions and runs each level's chunks in goroutines.

Classes (one opcode switch per step on the GPU; sub-lanes differ only in operands / small flags):
  M  products         OP_MUL, OP_MULC, OP_MULABC, OP_FMAC, OP_FMA (fused multiply-add, relin.py)
  X  boolean xor      OP_XORABC, OP_XOR
  A  linear           OP_ADD, OP_SUB, OP_ADDC, OP_NEG, OP_COPY, OP_SETC
  R  rows             OP_ABC
  I  inversions       OP_INV, OP_DIV
  BITS, BATCHINV      one instruction per step; the sub-lanes split its bits / its pairs
  HIST                one instruction per step: multiplicities of a lookup table (sub-lane 0 counts)
  COMMIT              one instruction per step: the solver stops, the prover commits (csrc/commit.hip)
List scheduling by longest path to a sink (class costs ~ instruction counts of the kernel).
"""
from __future__ import annotations

import heapq

from .api import (OP_ABC, OP_ADD, OP_ADDC, OP_BAND, OP_BATCHINV, OP_BITS, OP_BXOR, OP_COMMIT, OP_COPY,
                  OP_DIV, OP_EMUL, OP_HIST, OP_HQ, OP_INV, OP_MUL, OP_MULABC, OP_MULC, OP_NEG, OP_PAIR, OP_SETC,
                  OP_SUB, OP_XOR, OP_XORABC)

OP_FMAC, OP_FMA = 18, 19       # relin.py: (op, dst, x, const, addend) / (op, dst, x, y, addend)
CLS_M, CLS_X, CLS_A, CLS_R, CLS_I, CLS_BITS, CLS_BINV, CLS_HIST, CLS_COMMIT, CLS_B, CLS_EMUL, \
    CLS_LIMBS = range(1, 13)
# CLS_LIMBS: OP_BITS with at most LIMBS_MAX outputs (the range checker's limb hints): S of them share
# a step, one per sub-lane; a longer decomposition (ToBinary) is a CLS_BITS step of its own, its
# outputs split over the sub-lanes
LIMBS_MAX = 16
# CLS_B (byte-op hints) is a scheduling class of its own -- a step never mixes them with inversions --
# but runs in the kernel's CLS_I arm (its quads carry class CLS_I)
# the class field of an operand quad has three bits: HIST / COMMIT rows carry 0 there and their
# class in the header quad
SINGLE = (CLS_BITS, CLS_BINV, CLS_HIST, CLS_COMMIT, CLS_EMUL)
UNIT_HQ = (OP_HIST, OP_COMMIT, OP_EMUL)      # unit ops followed by o[2] OP_HQ operand rows
CLASS_OF = {OP_MUL: CLS_M, OP_MULC: CLS_M, OP_MULABC: CLS_M, OP_FMAC: CLS_M, OP_FMA: CLS_M, OP_XORABC: CLS_X, OP_XOR: CLS_X,
            OP_ADD: CLS_A, OP_SUB: CLS_A, OP_ADDC: CLS_A, OP_NEG: CLS_A, OP_COPY: CLS_A,
            OP_SETC: CLS_A, OP_ABC: CLS_R, OP_INV: CLS_I, OP_DIV: CLS_I, OP_BITS: CLS_BITS,
            OP_BATCHINV: CLS_BINV, OP_HIST: CLS_HIST, OP_COMMIT: CLS_COMMIT, OP_BXOR: CLS_B,
            OP_BAND: CLS_B, OP_EMUL: CLS_EMUL}
# relative time of one step of the class on a lone wavefront (instruction counts / 40)
COST = {CLS_M: 10, CLS_X: 12, CLS_A: 2, CLS_R: 3, CLS_I: 4000, CLS_BITS: 40, CLS_BINV: 6000,
        CLS_HIST: 4000, CLS_COMMIT: 20000, CLS_B: 20, CLS_EMUL: 150, CLS_LIMBS: 80}


def class_of(o):
    if o[0] == OP_BITS and (o[3] & 0xffff) <= LIMBS_MAX:
        return CLS_LIMBS
    return CLASS_OF[o[0]]


def n_rows_of(o):
    """operand rows that follow a unit op (OP_PAIR rows of OP_BATCHINV, OP_HQ rows of OP_HIST /
    OP_COMMIT)"""
    return o[1] if o[0] == OP_BATCHINV else o[2] if o[0] in UNIT_HQ else 0


def unit_outputs(o):
    """values a unit op with OP_HQ rows defines: the table's multiplicities (OP_HIST), the quotient
    and remainder limbs (OP_EMUL: low byte of b), the challenge (OP_COMMIT)"""
    op, dst, _a, b = o[:4]
    if op == OP_HIST:
        return range(dst, dst + b)
    if op == OP_EMUL:
        return range(dst, dst + (b & 0xff))
    return (dst,)


def bits_count(b):
    """values an OP_BITS with operand b defines (b = count | width << 16)"""
    return b & 0xffff


def reads_of(op, dst, a, b, z=None):
    if op == OP_FMA:
        return (a, b, z)
    if op == OP_FMAC:
        return (a, z)
    if op == OP_ABC:
        return (dst, a, b)
    if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULABC, OP_XORABC, OP_XOR, OP_BXOR, OP_BAND):
        return (a, b)
    if op in (OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_COPY, OP_BITS):
        return (a,)
    return ()


def schedule(ops, n_bits_vals, S):
    """ops: [(op, dst, a, b)] in a valid sequential order, SSA values (OP_BATCHINV followed by its
    OP_PAIR rows; OP_BITS defines values dst .. dst + b - 1 -- ``n_bits_vals`` maps its first
    value to the list of values it defines).  Returns a list of steps; a step is
    (class, [op index, ...]) with at most S indices (exactly one for BITS / BATCHINV)."""
    n = len(ops)
    producer = {}
    units = []          # schedulable units: op index (BATCHINV swallows its PAIR rows)
    i = 0
    while i < n:
        op, dst, a, b = ops[i][:4]
        units.append(i)
        if op == OP_BATCHINV:
            for k in range(1, dst + 1):
                producer[ops[i + k][1]] = i
            i += dst + 1
            continue
        if op in UNIT_HQ:
            for v in unit_outputs(ops[i]):
                producer[v] = i
            i += a + 1
            continue
        if op == OP_BITS:
            for v in n_bits_vals[dst]:
                producer[v] = i
        elif op != OP_ABC:
            producer[dst] = i
        i += 1
    succs = {u: [] for u in units}
    npred = {u: 0 for u in units}
    for u in units:
        op, dst, a, b = ops[u][:4]
        if op == OP_BATCHINV or op in UNIT_HQ:
            rd = [ops[u + k][2] for k in range(1, n_rows_of(ops[u]) + 1)]
        else:
            rd = reads_of(*ops[u])
        seen = set()
        for v in rd:
            p = producer.get(v)
            if p is not None and p != u and p not in seen:
                seen.add(p)
                succs[p].append(u)
                npred[u] += 1
    cls = {u: class_of(ops[u]) for u in units}
    # priority: longest weighted path to a sink
    prio = {}
    for u in reversed(units):
        best = 0
        for s_ in succs[u]:
            if prio[s_] > best:
                best = prio[s_]
        prio[u] = best + COST[cls[u]]
    ready = {c: [] for c in COST}
    for u in units:
        if npred[u] == 0:
            heapq.heappush(ready[cls[u]], (-prio[u], u))
    steps = []
    left = len(units)
    while left:
        # the class holding the most urgent ready operation goes next
        best_c, best_p = None, None
        for c, h in ready.items():
            if h and (best_p is None or h[0][0] < best_p):
                best_c, best_p = c, h[0][0]
        cap = 1 if best_c in SINGLE else S
        chosen = []
        h = ready[best_c]
        while h and len(chosen) < cap:
            chosen.append(heapq.heappop(h)[1])
        steps.append((best_c, chosen))
        left -= len(chosen)
        for u in chosen:
            for s_ in succs[u]:
                npred[s_] -= 1
                if npred[s_] == 0:
                    heapq.heappush(ready[cls[s_]], (-prio[s_], s_))
    return steps


def schedule_cost(steps):
    return sum(COST[c] for c, _ in steps)
