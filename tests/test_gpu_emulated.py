"""gnark ``std/math/emulated`` on the GPU: circuits over an emulated field are
solved by the device solver -- the product hint's big-integer division runs in the OP_EMUL arm of
solve_vliw_kernel<S, true> --, proved with the Groth16 commitment extension and compared bit for bit
(proof, commitment, proof of knowledge, per-proof status) with the C oracle, whose solver has its own
division (oracle/c/zkref_prove.inc, hint kind 7).  Reference users: hash/emulated/bn254/poseidon
(its test: poseidon_test.go:45-78), tree/smt/emulated.  Parity unpinned (tests/test_emulated.py)."""
import random

import pytest

from gnark_crypto_primitives_amd import circuits
from gnark_crypto_primitives_amd.frontend import compile_circuit
from gnark_crypto_primitives_amd.hash import poseidon_native
from gnark_crypto_primitives_amd.std import emulated as em
from tests import helpers as H
from tests.test_emulated import ArithCircuit, FormatCircuit, format_assignment
from tests.test_gpu_commitment import _check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("params,lanes", [(em.BN254Fr, 1), (em.Secp256k1Fp, 64), (em.BN254Fp, 8)],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_emulated_arithmetic(zk_ctx, params, lanes):
    circ = ArithCircuit(params)
    cc = compile_circuit(circ, lanes)
    p = params.modulus
    rng = random.Random(lanes)
    asg = [circ.assignment(rng.randrange(p), rng.randrange(p)) for _ in range(70)]
    asg[1] = circ.assignment(p - 1, 1)
    asg[2] = circ.assignment(0, 0)                  # x == y: flag 1, the select takes x
    x = rng.randrange(p)
    asg[3] = circ.assignment(x, x)
    asg[4] = circ.assignment(p - 1, p - 1)
    asg[9] = circ.assignment(x, 5, z=7)             # wrong product
    asg[63] = circ.assignment(x, 6, flag=1)
    asg[64] = circ.assignment(x, 7, s=x)            # wrong selected value
    bad = [9, 63, 64]
    limbs = list(asg[69]["X"])
    limbs[2] += 1 << 64                              # a witness limb wider than 64 bits
    asg[69] = dict(asg[69], X=limbs)
    bad.append(69)
    _check(zk_ctx, cc, asg, bad, 70 + lanes,
           publics=[list(a["Z"]) + [a["Flag"]] for a in asg])


def test_emulated_poseidon_matches_native(zk_ctx):
    """The reference's TestEmulatedPoseidonMatchesNative (inputs 1, 2, 3) and random inputs: 122 734
    constraints, 2^17 domain, ~8 * 10^2 product checks per proof."""
    cc = H.compiled("emulated-poseidon")
    assert cc.domain_log2() == 17 and len(cc.commitments) == 1
    mk = circuits.EmulatedPoseidonCircuit.assignment
    rng = random.Random(8)
    asg = [mk((1, 2, 3)), mk((0, 0, 0)), mk((H.R - 1, H.R - 2, 1)),
           mk([rng.randrange(H.R) for _ in range(3)]),
           mk((1, 2, 3), poseidon_native.hash([1, 2, 4]))]
    assert sum(v << (64 * i) for i, v in enumerate(asg[0]["Expected"])) == poseidon_native.hash([1, 2, 3])
    _check(zk_ctx, cc, asg, [4], 81, wbits=(0, 0), publics=[list(a["Expected"]) for a in asg],
           max_batch=64)


@pytest.mark.parametrize("to_te", [False, True], ids=["TEtoRTE", "RTEtoTE"])
def test_format_native_and_emulated(zk_ctx, to_te):
    """ecc/format/twistededwards_test.go:70-134 (both directions, native + emulated) proved on the GPU"""
    cc = compile_circuit(FormatCircuit(to_te))
    rng = random.Random(5)
    asg = [format_assignment(to_te)] + \
        [format_assignment(to_te, rng.randrange(H.R), rng.randrange(H.R)) for _ in range(66)]
    asg[17] = format_assignment(to_te, 3, 4, bad=True)
    _check(zk_ctx, cc, asg, [17], 90 + to_te)


def test_mimc7_native(zk_ctx):
    """iden3 MiMC7 (hash/native/bn254/mimc7/mimc_test.go:18-49): the reference's one-preimage circuit
    proved on the GPU and compared with the C oracle's proofs (the emulated variant runs through
    the interpreters in tests/test_mimc7.py; the emulated machinery on the GPU is the Poseidon test
    above)"""
    import numpy as np

    from gnark_crypto_primitives_amd import groth16
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
    from gnark_crypto_primitives_amd.hash import mimc7_native
    from oracle import cref
    from tests.test_mimc7 import _circuit
    rng = random.Random(12)
    cc = compile_circuit(_circuit(1))
    xs = [12] + [rng.randrange(H.R) for _ in range(69)]
    asg = [{"Hash": mimc7_native.hash([x]), "Preimages": [x]} for x in xs]
    asg[33]["Hash"] = (asg[33]["Hash"] + 1) % H.R
    pk, vk, _ = groth16.setup(cc, 3, groth16.gpu_mul(zk_ctx))
    prover = groth16.Prover(zk_ctx, cc, pk, 7, 5)
    inp = np.stack([to_mont_array(cc.assignment_vector(a)) for a in asg])
    rs = np.stack([to_mont_array([rng.randrange(H.R), rng.randrange(H.R)]) for _ in asg])
    proofs, status = prover.prove(inp, rs)
    prover.close()
    want, wstatus, _ = cref.groth16_prove_batch(cref.R1csHandle(cc), cref.PkHandle(pk), inp, rs)
    assert set(np.nonzero(status)[0]) == {33} == set(np.nonzero(wstatus)[0])
    ok = status == 0
    assert np.array_equal(proofs[ok], want[ok])


def test_cs_load_refuses_malformed_unit_rows(zk_ctx):
    """zkmi_cs_load checks every slot / constant / count of the unit rows on the host (OP_EMUL,
    packed limb steps, OP_HIST) before a kernel can use them as addresses."""
    import numpy as np

    from gnark_crypto_primitives_amd import lib
    from gnark_crypto_primitives_amd.frontend import schedule as sch
    from gnark_crypto_primitives_amd.frontend.compile import to_mont_array
    cc = compile_circuit(ArithCircuit(em.BN254Fr), 4)
    consts = to_mont_array(cc.consts)
    S = cc.lanes_per_proof

    def load(prog, n_consts=len(cc.consts), n_wires=cc.n_wires, n_slots=cc.v_n_slots):
        prog = np.ascontiguousarray(prog, dtype=np.uint32)
        cd = lib.CsDesc(n_wires, cc.n_public, cc.n_secret, cc.n_constraints, n_slots,
                        prog.shape[0], n_consts, S, prog.ctypes.data, consts.ctypes.data)
        h = zk_ctx.cs_load(cd)
        zk_ctx.cs_free(h)

    load(cc.vprogram)                                   # the real program loads
    hdr = cc.vprogram[:, 0, 0]
    emul = int(np.nonzero(hdr == sch.CLS_EMUL)[0][0])
    limbs = int(np.nonzero(hdr == sch.CLS_LIMBS)[0][0])
    hist = int(np.nonzero(hdr == sch.CLS_HIST)[0][0])

    def broken(row, quad, word, value):
        p = cc.vprogram.copy()
        p[row, quad, word] = value
        return p

    aux = int(cc.vprogram[emul, 0, 3])
    cases = [
        broken(emul, 1, 1, cc.n_wires - 2),                       # outputs run past the wires
        broken(emul, 0, 3, aux ^ 0x100),                          # header / quad disagree on aux
        broken(emul + 1, 1, 2, cc.v_n_slots),                     # operand slot out of range
        broken(emul, 0, 1, 9),                                    # operand count without its rows
        broken(limbs, 1, 2, cc.v_n_slots),                        # source slot out of range
        broken(limbs, 1, 3, 17 | 13 << 16),                       # more than 16 limbs in a packed step
        broken(limbs, 1, 3, 4 | 17 << 16),                        # limb width above 16
        broken(hist, 1, 1, cc.n_wires - 1),                       # counters run past the wires
        broken(hist + 1, 1, 2, cc.v_n_slots),                     # query slot out of range
    ]
    for i, p in enumerate(cases):
        with pytest.raises(lib.ZkmiError):
            load(p)
    p = cc.vprogram.copy()                                         # modulus constants past the pool
    p[emul, 0, 3] = p[emul, 1, 3] = (aux & 0xfff) | (len(cc.consts) - 3) << 12
    with pytest.raises(lib.ZkmiError):
        load(p)
    with pytest.raises(lib.ZkmiError):                            # sub-lane count not a power of two
        prog = np.ascontiguousarray(cc.vprogram, dtype=np.uint32)
        zk_ctx.cs_load(lib.CsDesc(cc.n_wires, cc.n_public, cc.n_secret, cc.n_constraints,
                                  cc.v_n_slots, prog.shape[0], len(cc.consts), 12,
                                  prog.ctypes.data, consts.ctypes.data))
