"""Host build of the solver's field inversion (csrc/modinv30.h, Bernstein-Yang divsteps on nine
30-bit limbs) against Python integers: both BN254 fields, edge values, random values."""
import os
import random
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_modinv30_host(tmp_path):
    exe = str(tmp_path / "test_modinv")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I",
                           os.path.join(ROOT, "gnark_crypto_primitives_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "test_modinv.cpp"), "-o", exe])
    rng = random.Random(30)
    cases = []
    for name, m in (("r", R), ("p", P)):
        xs = [0, 1, 2, 3, m - 1, m - 2, (m + 1) // 2, 1 << 253, (1 << 253) - 1, 1 << 30, (1 << 30) - 1,
              (1 << 60) + 1, m >> 1, 5, pow(5, (m - 1) // 4, m)]
        xs += [rng.randrange(m) for _ in range(400)]
        xs += [rng.randrange(1 << k) for k in (1, 8, 29, 31, 64, 128, 200) for _ in range(8)]
        cases += [(name, m, x) for x in xs]
    inp = "".join(f"{name} {x:064x}\n" for name, _, x in cases)
    out = subprocess.run([exe], input=inp, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [int(line, 16) for line in out.stdout.split()]
    assert len(got) == len(cases)
    for (name, m, x), y in zip(cases, got):
        want = pow(x, -1, m) if x else 0
        assert y == want, (name, hex(x), hex(y), hex(want))
