"""The C-ABI library loads and exports every symbol include/zkmi.h declares (no GPU needed)."""
import ctypes as C
import os
import re

from gnark_crypto_primitives_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "zkmi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zkmi_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 20
    handle = lib.load()
    bound = {n for n, _, _ in lib.SYMBOLS}
    for n in names:
        assert hasattr(handle, n), f"{n} declared in zkmi.h but missing from libzkmi.so"
        assert n in bound, f"{n} declared in zkmi.h but not bound in lib.py"
    assert bound <= set(names)


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of the descriptor structs as a C compiler lays out include/zkmi.h, against
    the ctypes mirrors in lib.py (what a cgo caller would see too)."""
    import subprocess
    from gnark_crypto_primitives_amd import plonk
    fields = {"zkmi_pk_desc": [n for n, _ in lib.PkDesc._fields_],
              "zkmi_commitment_desc": [n for n, _ in lib.CommitmentDesc._fields_],
              "zkmi_r1cs_desc": [n for n, _ in lib.R1csDesc._fields_],
              "zkmi_cs_desc": [n for n, _ in lib.CsDesc._fields_],
              "zkmi_plonk_pk_desc": [n for n, _ in plonk.PlonkDesc._fields_]}
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "zkmi.h"', 'int main(void) {']
    for st, fs in fields.items():
        prog.append(f'printf("{st} %zu\\n", sizeof({st}));')
        for f in fs:
            prog.append(f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    prog += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for st, cls in (("zkmi_pk_desc", lib.PkDesc), ("zkmi_cs_desc", lib.CsDesc),
                    ("zkmi_commitment_desc", lib.CommitmentDesc), ("zkmi_r1cs_desc", lib.R1csDesc),
                    ("zkmi_plonk_pk_desc", plonk.PlonkDesc)):
        assert int(out[st]) == C.sizeof(cls), st
        for f in fields[st]:
            assert int(out[f"{st}.{f}"]) == getattr(cls, f).offset, (st, f)


def test_no_cpu_fallback_without_gpu():
    """On a machine without a HIP device zkmi_init must refuse (ZKMI_ERR_NO_DEVICE), not fall back."""
    import torch
    if torch.cuda.is_available():
        return
    handle = lib.load()
    h = C.c_void_p()
    assert handle.zkmi_init(0, C.byref(h)) == -3
    try:
        lib.Context(0)
        raise AssertionError("Context() succeeded without a GPU")
    except lib.ZkmiError:
        pass
