"""Build-time guard for csrc/solve.hip: the solver issues its row stores from the fixed
accumulation registers a0..a31 in inline asm, after the next step's operand fetches; those stores
are invisible to the compiler's bookkeeping, so correctness needs that NOTHING else in
solve_vliw_kernel<S> lives in AGPRs -- no AGPR spill slots, no extra accumulation registers.  The
code-object remarks of every template instance must say: AGPRs = 32 exactly, no VGPR spills."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC) and not shutil.which("hipcc"), reason="hipcc missing")
def test_solver_kernels_keep_the_accumulation_registers_to_themselves():
    src = os.path.join(ROOT, "gnark_crypto_primitives_amd", "csrc", "solve.hip")
    err = subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "-O3", "-std=c++17",
                          "--offload-arch=gfx950", "--offload-device-only", "-mllvm",
                          "-pragma-unroll-threshold=1000000",
                          "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull],
                         capture_output=True, text=True, timeout=900).stderr
    seen, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"remark:\s+(AGPRs|VGPRs Spill): (\d+)", line)
        if m and cur and "solve_vliw_kernel" in cur:
            seen.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    # S sub-lanes x {without, with} the OP_EMUL arm (std/math/emulated product hints)
    lanes = sorted((int(m.group(1)), int(m.group(2)))
                   for m in (re.search(r"kernelILi(\d+)ELb(\d)E", k) for k in seen))
    assert lanes == [(s, e) for s in (1, 2, 4, 8, 16, 32, 64) for e in (0, 1)], lanes
    for name, r in seen.items():
        assert r == {"AGPRs": 32, "VGPRs Spill": 0}, (name, r)
