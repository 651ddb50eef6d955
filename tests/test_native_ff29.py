"""Host build of the GPU field/curve headers: the 9 x 29-bit lazy representation used by the MSM
kernels (csrc/ff29.h, ec29.h) against the reference representation (csrc/ff.h, ec.h)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_ff29_host_unit_tests(tmp_path):
    exe = str(tmp_path / "test_ff29")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I",
                           os.path.join(ROOT, "gnark_crypto_primitives_amd", "csrc"),
                           os.path.join(ROOT, "tests", "native", "test_ff29.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ff29 tests ok" in out.stdout
