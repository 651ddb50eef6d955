"""``frontend.Compile`` look-alike: circuit definition -> CompiledCircuit (R1CS + witness program).

In the reference every test enters gnark through
``frontend.Compile(ecc.BN254.ScalarField(), r1cs.NewBuilder, &circuit)``
(e.g. tree/test/verifier_bn254_test.go:41, hash/native/bn254/poseidon/poseidon_test.go:63) on a
struct whose ``frontend.Variable`` fields are the inputs (public ones tagged
``gnark:",public"``) and whose ``Define(api)`` builds the constraints.  Here a circuit is a class
with ``Public``/``Secret`` field declarations and a ``define(self, api)`` method.
"""
from __future__ import annotations

import hashlib

import numpy as np

from .api import (API, HINT_INVZERO, HINT_NBITS, OP_ABC, OP_ADD, OP_ADDC, OP_BAND, OP_BATCHINV,
                  OP_BITS, OP_BXOR, OP_COMMIT, OP_COPY, OP_DIV, OP_EMUL, OP_END, OP_HIST, OP_HQ, OP_INV, OP_MUL, OP_MULABC,
                  OP_MULC, OP_NEG, OP_PAIR, OP_SETC, OP_XOR, OP_XORABC,
                  OP_SUB, R)


class _Field:
    def __init__(self, n=None):
        self.n = n


class Public(_Field):
    """Public input (gnark struct tag ``gnark:",public"``); ``Public(n)`` declares an array."""


class Secret(_Field):
    """Secret input (gnark's default for an untagged frontend.Variable field)."""


def _fields(circuit):
    out = []
    for klass in reversed(type(circuit).__mro__):
        for name, f in vars(klass).items():
            if isinstance(f, _Field):
                out.append((name, f))
    return out


def int_to_limbs(x: int) -> np.ndarray:
    return np.frombuffer(int(x % (1 << 256)).to_bytes(32, "little"), dtype=np.uint64).copy()


def ints_to_array(xs) -> np.ndarray:
    """list of ints -> uint64 [len, 4] little-endian limbs."""
    buf = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).copy()


def array_to_ints(a: np.ndarray):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(a.shape[0])]


MONT_R = (1 << 256) % R
MONT_RINV = pow(MONT_R, R - 2, R)


def to_mont_array(xs) -> np.ndarray:
    return ints_to_array([x % R * MONT_R % R for x in xs])


def from_mont_array(a: np.ndarray):
    return [x * MONT_RINV % R for x in array_to_ints(a)]


SUB_LANE_CHOICES = (1, 2, 4, 8, 16, 32, 64)


def emul_unit(limbs, aux, consts):
    """What an OP_EMUL unit computes (api.py): limbs = the na limbs of a then the nb of b, aux =
    nout | na << 8 | first modulus constant << 12.  Returns the nout - 4 quotient limbs and the four
    remainder limbs (64 bits each)."""
    nout, na, c0 = aux & 0xff, (aux >> 8) & 0xf, aux >> 12
    a = sum(v << (64 * i) for i, v in enumerate(limbs[:na]))
    b = sum(v << (64 * i) for i, v in enumerate(limbs[na:]))
    p = sum(consts[c0 + i] << (64 * i) for i in range(4))
    k, rem = divmod(a * b, p)
    m = (1 << 64) - 1
    return [(k >> (64 * i)) & m for i in range(nout - 4)] + [(rem >> (64 * i)) & m for i in range(4)]


def build_vprogram(ops, val_wire, row_of, chk, n_wires, lanes_req=0):
    """Scheduled (VLIW) form of a witness program: the operations packed into steps of up to S
    independent operations of one class, S sub-lanes per proof (schedule.py).
    ops: SSA ops in a valid order; val_wire: value -> wire (its slot forever); row_of: op index ->
    constraint row it emits; chk: op index -> 1 when the solver must verify that row.
    Returns (vprogram uint32 [n_rows, 1 + S, 4], n_rows, n_steps, n_slots, S, cost); row = header
    quad (class, active, aux, 0) + S operand quads (op | chk << 5 | class << 6 | row_k << 9, dst
    slot, a, b: the class rides in every quad so that the kernel needs the header only for the
    two one-instruction classes); a
    BATCHINV step is followed by ceil(n / S) rows of (OP_PAIR, dst, src) quads.  Temporaries are
    recycled by step."""
    from . import schedule as sch
    bits_vals = {o[1]: range(o[1], o[1] + (o[3] & 0xffff)) for o in ops if o[0] == OP_BITS}
    best = None
    # auto: 4 sub-lanes (idle sub-lanes cost nothing: the solve is a latency chain on an
    # otherwise empty SIMD), more only while doubling them shortens the schedule by >= 8 %
    # (32 and 64 are accepted on request only: below four proofs per wavefront the value-file
    # accesses stop filling their 64-byte lines and the steps get slower than they get fewer --
    # measured)
    choices = (lanes_req,) if lanes_req else (4, 8, 16)
    for S in choices:
        if S not in SUB_LANE_CHOICES:
            raise ValueError(f"lanes_per_proof must be one of {SUB_LANE_CHOICES}")
        steps = sch.schedule(ops, bits_vals, S)
        cost = sch.schedule_cost(steps)
        if best is None or cost < 0.92 * best[1]:
            best = (S, cost, steps)
        else:
            break
    S, cost, steps = best
    # ---- last use (step index) of every value
    last = {}
    for t, (c, idxs) in enumerate(steps):
        for i in idxs:
            op, dst, a, b = ops[i][:4]
            if op == OP_BATCHINV or op in sch.UNIT_HQ:
                for q in range(1, sch.n_rows_of(ops[i]) + 1):
                    last[ops[i + q][2]] = t
            else:
                for v in sch.reads_of(*ops[i]):
                    last[v] = t
    slot = dict(val_wire)
    free, n_slots = [], n_wires
    rows = []

    def quad(i, c):
        op, dst, a, b = ops[i][:4]
        if op == sch.OP_FMA:       # d = a * b + z: the addend's slot rides in the row-index bits
            return (op | c << 6 | slot[ops[i][4]] << 9, slot[dst], slot[a], slot[b])
        if op == sch.OP_FMAC:      # d = a * const[b] + z
            return (op | c << 6 | slot[ops[i][4]] << 9, slot[dst], slot[a], b)
        if op == OP_ABC:
            return (op | chk.get(i, 0) << 5 | c << 6 | row_of[i] << 9, slot[dst], slot[a], slot[b])
        if c == sch.CLS_LIMBS:             # class bits 0: the kernel takes the class from the header
            return (op, slot[dst], slot[a], b)
        w0 = op | c << 6 | (row_of[i] << 9 if i in row_of else 0)
        if op in (OP_BXOR, OP_BAND):       # scheduled as CLS_B, executed by the kernel's CLS_I arm
            return (op | sch.CLS_I << 6, slot[dst], slot[a], slot[b])
        if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULABC, OP_XORABC, OP_XOR):
            return (w0, slot[dst], slot[a], slot[b])
        if op in (OP_MULC, OP_ADDC):
            return (w0, slot[dst], slot[a], b)
        if op in (OP_NEG, OP_INV, OP_COPY):
            return (w0, slot[dst], slot[a], 0)
        if op == OP_SETC:
            return (w0, slot[dst], 0, b)
        if op == OP_BITS:
            return (w0, slot[dst], slot[a], b)
        raise AssertionError(op)

    for t, (c, idxs) in enumerate(steps):
        released = []
        # destinations first get their slots (sources are all older values)
        for i in idxs:
            op, dst, a, b = ops[i][:4]
            if op == OP_BATCHINV:           # results of the unit: wires, or temporaries (api.py)
                for q in range(1, dst + 1):
                    d2 = ops[i + q][1]
                    if d2 not in slot:
                        if free:
                            slot[d2] = free.pop()
                        else:
                            slot[d2] = n_slots
                            n_slots += 1
                        if d2 not in last:
                            released.append(slot[d2])
                continue
            if op in (OP_ABC, OP_BITS) or op in sch.UNIT_HQ:
                continue
            if dst not in slot:
                if free:
                    slot[dst] = free.pop()
                else:
                    slot[dst] = n_slots
                    n_slots += 1
                if dst not in last:
                    released.append(slot[dst])
        hdr = (c, len(idxs), 0, 0)
        if c == sch.CLS_BINV:
            i = idxs[0]
            npairs = ops[i][1]
            nrows = -(-npairs // S)
            rows.append([(c, npairs, nrows, 0)] + [(c << 6, 0, 0, 0)] * S)
            pairs = [(OP_PAIR, slot[ops[i + q][1]], slot[ops[i + q][2]], 0)
                     for q in range(1, npairs + 1)]
            for r in range(nrows):
                chunk = pairs[r * S:(r + 1) * S]
                rows.append([(sch.CLS_BINV | 0x100, len(chunk), 0, 0)] + chunk +
                            [(0, 0, 0, 0)] * (S - len(chunk)))
        elif c in (sch.CLS_HIST, sch.CLS_COMMIT, sch.CLS_EMUL):
            # class field of the quads = 0: the kernel takes the class from the header quad.
            # HIST: header (class, n queries, n rows, table size), quad 0 = (OP_HIST, first slot);
            # then ceil(n / S) rows of (OP_HQ, 0, query slot, 0).  COMMIT: one row, header (class,
            # n operands, 0, commitment index), quad 0 = (OP_COMMIT, challenge slot, 0, index): the
            # host ends a kernel launch in front of it (zkmi_cs_load records the row).
            i = idxs[0]
            op, dst, n_q, aux = ops[i][:4]
            # EMUL: as HIST with quad 0 = (OP_EMUL, first slot, 0, nout | na << 8 | const << 12)
            if c in (sch.CLS_HIST, sch.CLS_EMUL):
                nrows = -(-n_q // S)
                rows.append([(c, n_q, nrows, aux), (op, slot[dst], 0, aux)] +
                            [(0, 0, 0, 0)] * (S - 1))
                qs = [(OP_HQ, 0, slot[ops[i + q][2]], 0) for q in range(1, n_q + 1)]
                for r in range(nrows):
                    chunk = qs[r * S:(r + 1) * S]
                    rows.append([(c | 0x100, len(chunk), 0, 0)] + chunk +
                                [(0, 0, 0, 0)] * (S - len(chunk)))
            else:
                rows.append([(c, n_q, 0, aux), (OP_COMMIT, slot[dst], 0, aux)] +
                            [(0, 0, 0, 0)] * (S - 1))
        else:
            kc = sch.CLS_I if c == sch.CLS_B else c       # class the kernel sees
            quads = [quad(i, c) for i in idxs]
            idle = (0, 0, 0, 0) if c == sch.CLS_LIMBS else (kc << 6, 0, 0, 0)
            rows.append([(kc, len(idxs), 0, 0)] + quads + [idle] * (S - len(quads)))
        # temporaries whose last reader is this step return to the pool for LATER steps
        for i in idxs:
            op, dst, a, b = ops[i][:4]
            srcs = [ops[i + q][2] for q in range(1, sch.n_rows_of(ops[i]) + 1)] \
                if (op == OP_BATCHINV or op in sch.UNIT_HQ) else sch.reads_of(*ops[i])
            for v in set(srcs):
                if last.get(v) == t and v not in val_wire and v in slot:
                    released.append(slot.pop(v))
        free.extend(released)
    if n_slots >= 1 << 23:
        raise ValueError("more than 2^23 value slots")     # FMA addends live in 23 bits
    vprogram = np.array(rows, dtype=np.uint32).reshape(len(rows), 1 + S, 4)
    return vprogram, len(rows), len(steps), n_slots, S, cost


class CompiledCircuit:
    """What gnark calls constraint.ConstraintSystem, for this framework.

    R1CS part (consumed by setup and by the oracle): CSR matrices L, R, O over ``n_wires`` columns,
    coefficient pool, solve order (instructions, hints).
    Program part (consumed by the GPU solver): uint32 [n_ops, 4] instructions over ``n_slots``
    value slots; slot i < n_wires *is* wire i.
    """

    def __init__(self, api: API, layout, lanes_per_proof=0, relinearize=True):
        self.layout = layout                      # [(name, n|None, public?)]
        self._lanes_req = lanes_per_proof
        self._relinearize = relinearize
        self.n_wires = api.n_wires
        self.n_public = api.n_public              # includes the ONE wire
        self.n_secret = api.n_secret
        self.n_constraints = len(api.constraints)
        self.n_inputs = api.n_public - 1 + api.n_secret
        # wires the builder knows to be boolean (bit decompositions, IsZero / And / Xor results):
        # the MSM table plan uses the fraction (groth16.Prover: zkmi_pk_desc.sparse_witness)
        self.boolean_wires = {key[0][0] for key in api.booleans
                              if len(key) == 1 and key[0][1] == 1 and key[0][0] != 0}
        self.n_boolean_wires = len(self.boolean_wires)
        self.consts = list(api.const_list)
        self.constraints = api.constraints
        self.instr = np.array(api.instr, dtype=np.uint32).reshape(-1, 2)
        self.hints = api.hints
        # Groth16 commitment extension (gnark constraint.Groth16Commitments): per commitment the
        # private committed wires (basis order), the hashed public / commitment wires and the wire
        # that receives the challenge
        self.commitments = [dict(c) for c in api.commitments]
        self.commit_fn = None      # run_program / run_vprogram: (index, hashed, committed) -> int
        self._build_csr(api)
        self._build_program(api)

    # ------------------------------------------------------------------ R1CS
    def _build_csr(self, api):
        cid = api._cid

        def csr(sel):
            ptr, col, c = [0], [], []
            for con in api.constraints:
                for w, v in sorted(con[sel].items()):
                    col.append(w)
                    c.append(cid(v))
                ptr.append(len(col))
            return (np.array(ptr, dtype=np.uint32), np.array(col, dtype=np.uint32),
                    np.array(c, dtype=np.uint32))

        self.L, self.Rm, self.O = csr(0), csr(1), csr(2)
        self.solve_wire = np.array([con[3] for con in api.constraints], dtype=np.int32)
        # hints: kind, inputs (each an LC), outputs
        kinds, in_ptr, lc_ptr, hcol, hcid, out_ptr, outs = [], [0], [0], [], [], [0], []
        for kind, ins, ows in api.hints:
            kinds.append(kind)
            for lc in ins:
                for w, v in sorted(lc.items()):
                    hcol.append(w)
                    hcid.append(cid(v))
                lc_ptr.append(len(hcol))
            in_ptr.append(len(lc_ptr) - 1)
            outs.extend(ows)
            out_ptr.append(len(outs))
        u32 = lambda x: np.array(x, dtype=np.uint32)
        self.hint_arrays = (u32(kinds), u32(in_ptr), u32(lc_ptr), u32(hcol), u32(hcid),
                            u32(out_ptr), u32(outs))
        self.consts = list(api.const_list)  # _cid may have appended

    # ------------------------------------------------------------------ witness program
    def _build_program(self, api):
        ops = api.ops
        n_vals = api.n_vals
        val_wire = api.val_wire
        # ---- dead code elimination: roots are constraint operands and wire-backed values
        live = np.zeros(n_vals + 1, dtype=bool)
        for v in val_wire:
            live[v] = True
        keep = [False] * len(ops)
        for i in range(len(ops) - 1, -1, -1):
            op, dst, a, b = ops[i]
            if op == OP_ABC:
                keep[i] = True
                live[dst] = live[a] = live[b] = True
            elif op in (OP_MULABC, OP_XORABC):       # they emit a row: always kept
                keep[i] = True
                live[a] = live[b] = True
            elif op == OP_BITS:
                keep[i] = True
                live[a] = True
            elif op in (OP_HIST, OP_COMMIT, OP_BATCHINV, OP_EMUL):
                keep[i] = True
            elif op in (OP_HQ, OP_PAIR):
                keep[i] = True
                live[a] = True
            elif live[dst]:
                keep[i] = True
                if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_BXOR, OP_BAND):
                    live[a] = live[b] = True
                elif op in (OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_COPY):
                    live[a] = True
        ops = [o for o, k in zip(ops, keep) if k]
        # ---- inversions of input wires (e.g. the IsZero of every SMT sibling) are hoisted to the
        # top and done with ONE field inversion (Montgomery's trick): OP_BATCHINV + (dst, src) rows
        n_in_wires = 1 + self.n_inputs
        hoist = [o for o in ops if o[0] == OP_INV and val_wire.get(o[2], n_in_wires) < n_in_wires
                 and o[1] in val_wire]
        if len(hoist) >= 4:
            hs = set(id(o) for o in hoist)
            ops = ([(OP_BATCHINV, len(hoist), 0, 0)] +
                   [(OP_PAIR, o[1], o[2], 0) for o in hoist] +
                   [o for o in ops if id(o) not in hs])
        # ---- last use of every value
        last = {}
        for i, (op, dst, a, b) in enumerate(ops):
            if op == OP_ABC:
                last[dst] = last[a] = last[b] = i
            elif op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULABC, OP_XORABC, OP_BXOR, OP_BAND):
                last[a] = last[b] = i
            elif op in (OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_COPY, OP_BITS, OP_PAIR, OP_HQ):
                last[a] = i
        # ---- slot assignment: wire-backed values live in their wire's slot forever; the rest
        # share a pool of temporaries above n_wires, recycled after the last use
        slot = dict(val_wire)
        free, n_slots = [], self.n_wires
        prog = np.zeros((len(ops) + 1, 4), dtype=np.uint32)
        n_abc = 0
        for i, (op, dst, a, b) in enumerate(ops):
            srcs = ()
            if op == OP_BATCHINV:
                prog[i] = (op, dst, 0, 0)           # dst field = number of (dst, src) rows
                continue
            if op == OP_PAIR:
                if dst not in slot:                     # a temporary result (api._batch_inverse_vals)
                    if free:
                        slot[dst] = free.pop()
                    else:
                        slot[dst] = n_slots
                        n_slots += 1
                prog[i] = (op, slot[dst], slot[a], 0)
                continue
            if op in (OP_HIST, OP_COMMIT, OP_EMUL):
                prog[i] = (op, slot[dst], a, b)         # dst wire-backed; a operand rows follow
                continue
            if op == OP_HQ:
                prog[i] = (op, 0, slot[a], 0)
                if last.get(a) == i and a not in val_wire:
                    free.append(slot[a])
                continue
            if op == OP_ABC:
                srcs = (dst, a, b)
                # bit 8: the solver verifies this row (assertion / division)
                flag = 0x100 if api.constraints[n_abc][5] else 0
                n_abc += 1
                prog[i] = (op | flag, slot[dst], slot[a], slot[b])
            elif op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULABC, OP_XORABC, OP_BXOR, OP_BAND):
                srcs = (a, b)
                if op in (OP_MULABC, OP_XORABC):
                    n_abc += 1
            elif op in (OP_MULC, OP_ADDC, OP_NEG, OP_INV, OP_COPY, OP_BITS):
                srcs = (a,)
            sa = slot[a] if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULC, OP_ADDC, OP_NEG,
                                   OP_INV, OP_COPY, OP_BITS, OP_MULABC, OP_XORABC, OP_BXOR,
                                   OP_BAND) else 0
            sb = slot[b] if op in (OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_MULABC, OP_XORABC, OP_BXOR,
                                   OP_BAND) else b
            for s in set(srcs):
                if last.get(s) == i and s not in val_wire:
                    free.append(slot[s])
            if op != OP_ABC:
                if dst not in slot:
                    if free:
                        slot[dst] = free.pop()
                    else:
                        slot[dst] = n_slots
                        n_slots += 1
                    if dst not in last:          # defined, never read (cannot happen after DCE)
                        free.append(slot[dst])
                prog[i] = (op, slot[dst], sa, sb)
        prog[len(ops)] = (OP_END, 0, 0, 0)
        self.program = prog
        self.n_slots = n_slots
        self.n_ops = len(ops)
        self._build_vprogram(api, ops)

    # ------------------------------------------------------------------ VLIW witness program
    def _build_vprogram(self, api, ops):
        """The program the GPU solver runs (csrc/solve.hip): see ``build_vprogram``."""
        self._ops = ops                       # post-DCE SSA ops: the PLONK lowering starts here
        self._val_wire = dict(api.val_wire)
        if self._relinearize:
            from .relin import relinearize
            ops, _ = relinearize(ops, api.val_wire, self.consts, api.n_vals)
        self.n_vops = len(ops)
        row_of, k = {}, 0                     # constraint row of every row-emitting op
        for i, o in enumerate(ops):
            if o[0] in (OP_ABC, OP_MULABC, OP_XORABC):
                row_of[i] = k
                k += 1
        chk = {i: (1 if api.constraints[r][5] else 0) for i, r in row_of.items()
               if ops[i][0] == OP_ABC}
        self._row_check = [bool(api.constraints[r][5]) for r in range(k)]
        (self.vprogram, self.v_n_rows, self.v_n_steps, self.v_n_slots, self.lanes_per_proof,
         self.schedule_cost) = build_vprogram(ops, api.val_wire, row_of, chk, self.n_wires,
                                              self._lanes_req)

    def run_vprogram(self, inputs):
        """Evaluate ``vprogram`` with Python integers exactly as the GPU kernel does (all reads of
        a step before its writes; sub-lanes in index order) -- the CPU check of the scheduler and
        of the step-wise slot recycling.  Returns (wires, a, b, c)."""
        from . import schedule as sch
        if len(inputs) != self.n_inputs:
            raise ValueError(f"expected {self.n_inputs} inputs, got {len(inputs)}")
        s = [0] * self.v_n_slots
        s[0] = 1
        for i, v in enumerate(inputs):
            s[1 + i] = int(v) % R
        nc = self.n_constraints
        a_, b_, c_ = [None] * nc, [None] * nc, [None] * nc
        C = self.consts
        self.last_status = 0
        rows = self.vprogram.tolist()
        r = 0
        while r < len(rows):
            hdr, quads = rows[r][0], rows[r][1:]
            cls = hdr[0] & 0xff
            r += 1
            if cls == sch.CLS_BINV:
                for _ in range(hdr[2]):
                    for w0, d, x, _y in rows[r][1:]:
                        if w0 & 0x1f == OP_PAIR:
                            s[d] = pow(s[x], R - 2, R)
                    r += 1
                continue
            if cls == sch.CLS_HIST:
                first, size = quads[0][1], hdr[3]
                for j in range(size):
                    s[first + j] = 0
                for _ in range(hdr[2]):
                    for w0, d, x, _y in rows[r][1:]:
                        if w0 & 0x1f == OP_HQ and s[x] < size:
                            s[first + s[x]] += 1
                    r += 1
                continue
            if cls == sch.CLS_EMUL:
                first, aux, ins = quads[0][1], hdr[3], []
                for _ in range(hdr[2]):
                    ins.extend(s[x] for w0, d, x, _y in rows[r][1:] if w0 & 0x1f == OP_HQ)
                    r += 1
                for j, v in enumerate(emul_unit(ins, aux, C)):
                    s[first + j] = v
                continue
            if cls == sch.CLS_COMMIT:
                s[quads[0][1]] = self._commit_value(hdr[3], s)
                continue
            writes = []
            for w0, d, x, y in quads:
                op, chk, k = w0 & 0x1f, (w0 >> 5) & 1, w0 >> 9
                if op == OP_END:
                    continue
                if op == OP_MUL:
                    writes.append((d, s[x] * s[y] % R))
                elif op == sch.OP_FMA:
                    writes.append((d, (s[x] * s[y] + s[k]) % R))
                elif op == sch.OP_FMAC:
                    writes.append((d, (s[x] * C[y] + s[k]) % R))
                elif op == OP_MULC:
                    writes.append((d, s[x] * C[y] % R))
                elif op == OP_MULABC:
                    v = s[x] * s[y] % R
                    a_[k], b_[k], c_[k] = s[x], s[y], v
                    writes.append((d, v))
                elif op == OP_XORABC:
                    ab2 = 2 * s[x] * s[y] % R
                    a_[k], b_[k], c_[k] = 2 * s[x] % R, s[y], ab2
                    writes.append((d, (s[x] + s[y] - ab2) % R))
                elif op == OP_XOR:
                    writes.append((d, (s[x] + s[y] - 2 * s[x] * s[y]) % R))
                elif op == OP_ADD:
                    writes.append((d, (s[x] + s[y]) % R))
                elif op == OP_SUB:
                    writes.append((d, (s[x] - s[y]) % R))
                elif op == OP_ADDC:
                    writes.append((d, (s[x] + C[y]) % R))
                elif op == OP_NEG:
                    writes.append((d, (-s[x]) % R))
                elif op == OP_COPY:
                    writes.append((d, s[x]))
                elif op == OP_SETC:
                    writes.append((d, C[y]))
                elif op == OP_INV:
                    writes.append((d, pow(s[x], R - 2, R)))
                elif op == OP_DIV:
                    writes.append((d, s[x] * pow(s[y], R - 2, R) % R))
                elif op in (OP_BXOR, OP_BAND):       # low 32 bits, as the kernel
                    u, v = s[x] & 0xffffffff, s[y] & 0xffffffff
                    writes.append((d, u ^ v if op == OP_BXOR else u & v))
                elif op == OP_ABC:
                    a_[k], b_[k], c_[k] = s[d], s[x], s[y]
                    if chk and s[d] * s[x] % R != s[y]:
                        self.last_status = -5
                elif op == OP_BITS:
                    v, width = s[x], (y >> 16) or 1
                    for j in range(y & 0xffff):
                        writes.append((d + j, (v >> (j * width)) & ((1 << width) - 1)))
                else:
                    raise AssertionError(op)
            for d, v in writes:
                s[d] = v
        return s[:self.n_wires], a_, b_, c_

    # ------------------------------------------------------------------ CPU evaluation
    def run_program(self, inputs):
        """Evaluate the witness program with Python integers (debug/test engine only; the product
        path runs csrc/solve.hip).  Returns (wires, a, b, c) as lists of ints."""
        if len(inputs) != self.n_inputs:
            raise ValueError(f"expected {self.n_inputs} inputs, got {len(inputs)}")
        s = [0] * self.n_slots
        s[0] = 1
        for i, v in enumerate(inputs):
            s[1 + i] = int(v) % R
        a_, b_, c_ = [], [], []
        C = self.consts
        self.last_status = 0
        hist = emul = None
        for op, d, a, b in self.program.tolist():
            chk, op = op & 0x100, op & 0xff
            if op not in (OP_HQ, OP_HIST):
                hist = None
            if op not in (OP_HQ, OP_EMUL) and emul is not None:
                emul = None
            if op == OP_MUL:
                s[d] = s[a] * s[b] % R
            elif op == OP_ADD:
                s[d] = (s[a] + s[b]) % R
            elif op == OP_SUB:
                s[d] = (s[a] - s[b]) % R
            elif op == OP_MULC:
                s[d] = s[a] * C[b] % R
            elif op == OP_ADDC:
                s[d] = (s[a] + C[b]) % R
            elif op == OP_ABC:
                a_.append(s[d])
                b_.append(s[a])
                c_.append(s[b])
                if chk and s[d] * s[a] % R != s[b]:
                    self.last_status = -5
            elif op == OP_MULABC:
                a_.append(s[a])
                b_.append(s[b])
                s[d] = s[a] * s[b] % R
                c_.append(s[d])
            elif op == OP_XORABC:
                ab2 = 2 * s[a] * s[b] % R
                a_.append(2 * s[a] % R)
                b_.append(s[b])
                c_.append(ab2)
                s[d] = (s[a] + s[b] - ab2) % R
            elif op == OP_NEG:
                s[d] = (-s[a]) % R
            elif op == OP_SETC:
                s[d] = C[b]
            elif op == OP_INV:
                s[d] = pow(s[a], R - 2, R)
            elif op == OP_DIV:
                s[d] = s[a] * pow(s[b], R - 2, R) % R
            elif op in (OP_BXOR, OP_BAND):
                u, v = s[a] & 0xffffffff, s[b] & 0xffffffff
                s[d] = u ^ v if op == OP_BXOR else u & v
            elif op == OP_BITS:
                v, width = s[a], (b >> 16) or 1
                for k in range(b & 0xffff):
                    s[d + k] = (v >> (k * width)) & ((1 << width) - 1)
            elif op == OP_HIST:
                hist = (d, b)
                for k in range(b):
                    s[d + k] = 0
            elif op == OP_EMUL:
                emul = [d, b, a, []]                  # first slot, aux, rows still to come, limbs
            elif op == OP_HQ and emul is not None:
                emul[3].append(s[a])
                emul[2] -= 1
                if emul[2] == 0:
                    for k, v in enumerate(emul_unit(emul[3], emul[1], C)):
                        s[emul[0] + k] = v
                    emul = None
            elif op == OP_HQ:
                if hist is not None and s[a] < hist[1]:     # rows of an OP_COMMIT only order it
                    s[hist[0] + s[a]] += 1
            elif op == OP_COMMIT:
                hist = None
                s[d] = self._commit_value(b, s)
            elif op == OP_COPY:
                s[d] = s[a]
            elif op == OP_PAIR:                      # row of a preceding OP_BATCHINV
                s[d] = pow(s[a], R - 2, R)
            elif op == OP_END:
                break
        return s[:self.n_wires], a_, b_, c_

    def _commit_value(self, idx, s):
        """challenge of commitment ``idx`` from the current slot values: ``commit_fn(idx, hashed
        values, committed values)`` -- groth16.commit_fn(pk) is the real one (Pedersen MSM +
        hash-to-field); without one a SHA-256 stand-in keeps the CPU interpreters self-contained
        (the circuit logic does not depend on where the challenge comes from)."""
        c = self.commitments[idx]
        hashed = [s[w] for w in c["hashed"]]
        committed = [s[w] for w in c["private"]]
        if self.commit_fn is not None:
            return int(self.commit_fn(idx, hashed, committed)) % R
        h = hashlib.sha256(b"stand-in commitment %d" % idx)
        for v in hashed + committed:
            h.update(int(v).to_bytes(32, "big"))
        return int.from_bytes(h.digest(), "big") % R

    def is_satisfied(self, wires):
        """Check every constraint <L,w>*<R,w> == <O,w> on a full wire assignment."""
        for k, (L, Rr, O, *_) in enumerate(self.constraints):
            ev = lambda lc: sum(c * wires[w] for w, c in lc.items()) % R
            if ev(L) * ev(Rr) % R != ev(O):
                return False, k
        return True, -1

    def domain_log2(self):
        n = max(self.n_constraints, 2)
        return (n - 1).bit_length()

    def assignment_vector(self, assignment: dict):
        """dict field name -> int | list[int]  ->  flat input vector in wire order."""
        out = []
        for name, n, _ in self.layout:
            v = assignment[name]
            if n is None:
                out.append(int(v) % R)
            else:
                if len(v) != n:
                    raise ValueError(f"{name}: expected {n} values")
                out.extend(int(x) % R for x in v)
        return out

    def fingerprint(self) -> str:
        """Identity of the constraint system (matrices and wire layout) -- not of the witness
        program, which may be re-scheduled without changing any proof."""
        h = hashlib.sha256()
        for arr in (*self.L, *self.Rm, *self.O):
            h.update(np.ascontiguousarray(arr).tobytes())
        h.update(repr((self.n_wires, self.n_public, self.n_secret)).encode())
        return h.hexdigest()[:16]


def compile_circuit(circuit, lanes_per_proof: int = 0, relinearize: bool = True) -> CompiledCircuit:
    """``frontend.Compile(field, r1cs.NewBuilder, circuit)`` for BN254's scalar field.
    lanes_per_proof: sub-lanes of the GPU solver per proof (1, 2, 4, ... 64; 0 = chosen from the
    schedule lengths)."""
    api = API()
    fields = _fields(circuit)
    layout = []
    # gnark wire order: ONE, public..., secret..., internal...
    for want_public in (True, False):
        for name, f in fields:
            if isinstance(f, Public) != want_public:
                continue
            mk = api.public_input if want_public else api.secret_input
            if f.n is None:
                setattr(circuit, name, mk(name))
            else:
                setattr(circuit, name, [mk(f"{name}[{i}]") for i in range(f.n)])
            layout.append((name, f.n, want_public))
    circuit.define(api)
    api.finalize()                    # deferred builders: range checker, multi-commitment
    return CompiledCircuit(api, layout, lanes_per_proof, relinearize)
