"""Every bench workload's synthetic assignments satisfy their circuit (CPU engine)."""
import random

import pytest

from gnark_crypto_primitives_amd import workloads
from gnark_crypto_primitives_amd.frontend import compile_circuit


@pytest.mark.parametrize("name", [n for n in workloads.NAMES if n not in ("address", "address-bytes", "emulated-poseidon")])
def test_workload_assignments_satisfy(name):
    circuit, gen, label = workloads.build(name, levels=12, populated=4)
    cc = compile_circuit(circuit)
    rng = random.Random(21)
    seen = set()
    for _ in range(4):
        a = gen(rng)
        wires = cc.run_program(cc.assignment_vector(a))[0]
        assert cc.is_satisfied(wires)[0] and cc.last_status == 0
        if "Fnc" in a:
            seen.add(a["Fnc"])
    if name == "verifier":
        assert seen == {0, 1}


def test_unknown_workload():
    with pytest.raises(ValueError):
        workloads.build("plonk")
