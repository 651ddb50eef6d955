"""ctypes binding of libzkmi.so (include/zkmi.h).  Loading fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKMI_LIB", os.path.join(_HERE, "libzkmi.so"))

ZKMI_OK = 0
ZKMI_ERR_UNSATISFIED = -5


class ZkmiError(RuntimeError):
    pass


class PkDesc(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("n_wires", C.c_uint32), ("n_a", C.c_uint32),
                ("n_b", C.c_uint32), ("n_k", C.c_uint32), ("n_z", C.c_uint32),
                ("a_wire", C.c_void_p), ("b_wire", C.c_void_p), ("k_wire", C.c_void_p),
                ("g1_a", C.c_void_p), ("g1_b", C.c_void_p), ("g1_k", C.c_void_p),
                ("g1_z", C.c_void_p), ("g2_b", C.c_void_p), ("g1_alpha", C.c_void_p),
                ("g1_beta", C.c_void_p), ("g1_delta", C.c_void_p), ("g2_beta", C.c_void_p),
                ("g2_delta", C.c_void_p), ("window_bits_g1", C.c_uint32),
                ("window_bits_g2", C.c_uint32),
                ("infinity_a", C.c_void_p), ("infinity_b", C.c_void_p), ("n_public", C.c_uint32),
                ("max_batch", C.c_uint32), ("table_budget_bytes", C.c_uint64),
                ("n_slots_hint", C.c_uint32), ("msm_chunk_factor", C.c_uint32),
                ("sparse_witness", C.c_uint32), ("n_commitments", C.c_uint32),
                ("commitments", C.c_void_p)]


class CommitmentDesc(C.Structure):
    _fields_ = [("n_private", C.c_uint32), ("n_hashed", C.c_uint32),
                ("commitment_wire", C.c_uint32), ("reserved", C.c_uint32),
                ("private_wires", C.c_void_p), ("hashed_wires", C.c_void_p),
                ("basis", C.c_void_p), ("basis_exp_sigma", C.c_void_p)]


class CsDesc(C.Structure):
    _fields_ = [("n_wires", C.c_uint32), ("n_public", C.c_uint32), ("n_secret", C.c_uint32),
                ("n_constraints", C.c_uint32), ("n_slots", C.c_uint32), ("n_rows", C.c_uint32),
                ("n_consts", C.c_uint32), ("lanes_per_proof", C.c_uint32), ("program", C.c_void_p),
                ("consts", C.c_void_p)]


class R1csDesc(C.Structure):
    _fields_ = [("n_wires", C.c_uint32), ("n_constraints", C.c_uint32), ("n_coeffs", C.c_uint32),
                ("coeffs", C.c_void_p), ("l_ptr", C.c_void_p), ("l_terms", C.c_void_p),
                ("r_ptr", C.c_void_p), ("r_terms", C.c_void_p), ("o_ptr", C.c_void_p),
                ("o_terms", C.c_void_p)]


# every symbol include/zkmi.h declares: (name, restype, argtypes)
_P, _SZ, _I = C.c_void_p, C.c_size_t, C.c_int
SYMBOLS = [
    ("zkmi_init", _I, [_I, C.POINTER(_P)]),
    ("zkmi_destroy", None, [_P]),
    ("zkmi_last_error", C.c_char_p, [_P]),
    ("zkmi_sync", _I, [_P]),
    ("zkmi_stream", _P, [_P]),
    ("zkmi_field_mul", _I, [_P, _I, _P, _P, _P, _SZ]),
    ("zkmi_field_mul_bench", _I, [_P, _I, _SZ, _I, C.POINTER(C.c_double)]),
    ("zkmi_ntt_batch", _I, [_P, _P, _I, _SZ, _I, _I]),
    ("zkmi_h_batch", _I, [_P, _P, _P, _P, _P, _I, _SZ]),
    ("zkmi_msm_bases_load", _I, [_P, _I, _P, _SZ, _I, C.POINTER(_P)]),
    ("zkmi_msm_bases_free", None, [_P, _P]),
    ("zkmi_msm_batch", _I, [_P, _P, _P, _SZ, _P]),
    ("zkmi_fixed_base_mul", _I, [_P, _I, _P, _P, _SZ, _P]),
    ("zkmi_pk_load", _I, [_P, C.POINTER(PkDesc), C.POINTER(_P)]),
    ("zkmi_pk_free", None, [_P, _P]),
    ("zkmi_pk_info", _I, [_P, C.POINTER(C.c_uint64)]),
    ("zkmi_cs_load", _I, [_P, C.POINTER(CsDesc), C.POINTER(_P)]),
    ("zkmi_cs_free", None, [_P, _P]),
    ("zkmi_solve_batch", _I, [_P, _P, _P, _SZ, _P, _P, _P]),
    ("zkmi_prove_batch", _I, [_P, _P, _P, _P, _SZ, _P, _P, _P]),
    ("zkmi_prove_submit", _I, [_P, _P, _P, _P, _SZ, _P]),
    ("zkmi_prove_collect", _I, [_P, _P, _P]),
    ("zkmi_prove_collect_ex", _I, [_P, _P, _P, _P]),
    ("zkmi_prove_witness_batch", _I, [_P, _P, _P, _P, _P, _P, _SZ, _SZ, _P, _P]),
    ("zkmi_host_alloc", _P, [_P, _SZ]),
    ("zkmi_host_free", None, [_P, _P]),
    ("zkmi_set_copy_threads", _I, [_P, _I]),
    ("zkmi_r1cs_load", _I, [_P, C.POINTER(R1csDesc), C.POINTER(_P)]),
    ("zkmi_r1cs_free", None, [_P, _P]),
    ("zkmi_prove_witness_submit", _I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _SZ, _P]),
    ("zkmi_last_timings", _I, [_P, C.POINTER(C.c_double)]),
    ("zkmi_plonk_pk_load", _I, [_P, _P, C.POINTER(_P)]),
    ("zkmi_plonk_pk_free", None, [_P, _P]),
    ("zkmi_plonk_round1", _I, [_P, _P, _P, _P, _SZ, _P, _P, _P]),
    ("zkmi_plonk_round2", _I, [_P, _P, _P, _P]),
    ("zkmi_plonk_round3", _I, [_P, _P, _P, _P]),
    ("zkmi_plonk_round4", _I, [_P, _P, _P, _P]),
    ("zkmi_plonk_round5", _I, [_P, _P, _P, _P]),
    ("zkmi_plonk_prove", _I, [_P, _P, _P, _P, _SZ, _P, _P, _P, _P]),
]

_lib = None


def load():
    """dlopen libzkmi.so and bind every declared symbol; raises if the library is missing.
    A process that also uses torch on the GPU must import torch BEFORE this call: torch ships its
    own libamdhip64.so.7, and the HIP runtime loaded first serves every later library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkmiError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as "
                        "g; g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(x):
    """numpy array / torch tensor / int -> raw address (host or device)."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("array must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr()
    return int(x)


class Context:
    """One per GPU (zkmi_init / zkmi_destroy)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = C.c_void_p()
        rc = self.lib.zkmi_init(device, C.byref(h))
        if rc != 0:
            raise ZkmiError(f"zkmi_init(device={device}) failed with {rc} "
                            "(-3 = no HIP device: this library only runs on a GPU)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkmi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.zkmi_last_error(self.h)
            raise ZkmiError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def sync(self):
        self._check(self.lib.zkmi_sync(self.h), "zkmi_sync")

    def stream(self) -> int:
        return self.lib.zkmi_stream(self.h) or 0

    # ---- primitives
    def field_mul(self, which, a, b, out=None):
        n = a.shape[0]
        out = np.empty_like(a) if out is None else out
        self._check(self.lib.zkmi_field_mul(self.h, which, _ptr(a), _ptr(b), _ptr(out), n),
                    "zkmi_field_mul")
        return out

    def field_mul_bench(self, which=1, n_threads=1 << 20, iters=256) -> float:
        r = C.c_double()
        self._check(self.lib.zkmi_field_mul_bench(self.h, which, n_threads, iters, C.byref(r)),
                    "zkmi_field_mul_bench")
        return r.value

    def ntt_batch(self, data, log_n, batch, inverse=False, coset=False):
        self._check(self.lib.zkmi_ntt_batch(self.h, _ptr(data), log_n, batch, int(inverse),
                                            int(coset)), "zkmi_ntt_batch")
        return data

    def h_batch(self, a, b, c, out, log_n, batch):
        self._check(self.lib.zkmi_h_batch(self.h, _ptr(a), _ptr(b), _ptr(c), _ptr(out), log_n,
                                          batch), "zkmi_h_batch")
        return out

    def msm_bases_load(self, group, bases, n, window_bits=0):
        h = C.c_void_p()
        self._check(self.lib.zkmi_msm_bases_load(self.h, group, _ptr(bases), n, window_bits,
                                                 C.byref(h)), "zkmi_msm_bases_load")
        return h

    def msm_bases_free(self, h):
        self.lib.zkmi_msm_bases_free(self.h, h)

    def msm_batch(self, bases_h, scalars, batch, out):
        self._check(self.lib.zkmi_msm_batch(self.h, bases_h, _ptr(scalars), batch, _ptr(out)),
                    "zkmi_msm_batch")
        return out

    def fixed_base_mul(self, group, base, scalars, n, out):
        self._check(self.lib.zkmi_fixed_base_mul(self.h, group, _ptr(base), _ptr(scalars), n,
                                                 _ptr(out)), "zkmi_fixed_base_mul")
        return out

    # ---- key / system handles
    def pk_load(self, desc: PkDesc):
        h = C.c_void_p()
        self._check(self.lib.zkmi_pk_load(self.h, C.byref(desc), C.byref(h)), "zkmi_pk_load")
        return h

    def pk_info(self, h):
        arr = (C.c_uint64 * 10)()
        self._check(self.lib.zkmi_pk_info(h, arr), "zkmi_pk_info")
        return dict(zip(("g1_windows", "g1_entries_per_base", "g2_windows", "g2_entries_per_base",
                         "g1_table_bytes", "g2_table_bytes", "g1_shared", "g2_shared",
                         "g1_comb_k", "g2_comb_k"), [int(x) for x in arr]))

    def pk_free(self, h):
        self.lib.zkmi_pk_free(self.h, h)

    def cs_load(self, desc: CsDesc):
        h = C.c_void_p()
        self._check(self.lib.zkmi_cs_load(self.h, C.byref(desc), C.byref(h)), "zkmi_cs_load")
        return h

    def cs_free(self, h):
        self.lib.zkmi_cs_free(self.h, h)

    def solve_batch(self, cs_h, inputs, batch, wires_out=None, abc_out=None, status_out=None):
        if status_out is None:
            status_out = np.zeros(batch, dtype=np.int32)
        self._check(self.lib.zkmi_solve_batch(self.h, cs_h, _ptr(inputs), batch, _ptr(wires_out),
                                              _ptr(abc_out), _ptr(status_out)), "zkmi_solve_batch")
        return status_out

    def prove_batch(self, pk_h, cs_h, inputs, batch, rs, proofs_out, status_out):
        self._check(self.lib.zkmi_prove_batch(self.h, pk_h, cs_h, _ptr(inputs), batch, _ptr(rs),
                                              _ptr(proofs_out), _ptr(status_out)),
                    "zkmi_prove_batch")

    def prove_submit(self, pk_h, cs_h, inputs, batch, rs):
        self._check(self.lib.zkmi_prove_submit(self.h, pk_h, cs_h, _ptr(inputs), batch, _ptr(rs)),
                    "zkmi_prove_submit")

    def prove_collect(self, proofs_out, status_out, commitments_out=None):
        if commitments_out is None:
            self._check(self.lib.zkmi_prove_collect(self.h, _ptr(proofs_out), _ptr(status_out)),
                        "zkmi_prove_collect")
        else:
            self._check(self.lib.zkmi_prove_collect_ex(self.h, _ptr(proofs_out), _ptr(status_out),
                                                       _ptr(commitments_out)),
                        "zkmi_prove_collect_ex")

    def prove_witness_batch(self, pk_h, wires, a, b, c, n_constraints, batch, rs, proofs_out):
        self._check(self.lib.zkmi_prove_witness_batch(self.h, pk_h, _ptr(wires), _ptr(a), _ptr(b),
                                                      _ptr(c), n_constraints, batch, _ptr(rs),
                                                      _ptr(proofs_out)),
                    "zkmi_prove_witness_batch")

    def prove_witness_submit(self, pk_h, r1cs_h, wires, a, b, c, n_constraints, batch, rs):
        self._check(self.lib.zkmi_prove_witness_submit(self.h, pk_h, r1cs_h, _ptr(wires), _ptr(a),
                                                       _ptr(b), _ptr(c), n_constraints, batch,
                                                       _ptr(rs)), "zkmi_prove_witness_submit")

    def r1cs_load(self, desc: "R1csDesc"):
        h = C.c_void_p()
        self._check(self.lib.zkmi_r1cs_load(self.h, C.byref(desc), C.byref(h)), "zkmi_r1cs_load")
        return h

    def r1cs_free(self, h):
        self.lib.zkmi_r1cs_free(self.h, h)

    def host_alloc(self, shape, dtype=np.uint64):
        """numpy array over page-locked memory from zkmi_host_alloc (free with host_free)."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.lib.zkmi_host_alloc(self.h, nbytes)
        if not p:
            raise ZkmiError(f"zkmi_host_alloc({nbytes}) failed")
        buf = (C.c_uint8 * nbytes).from_address(p)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.zkmi_host_free(self.h, p)

    def set_copy_threads(self, n):
        self._check(self.lib.zkmi_set_copy_threads(self.h, n), "zkmi_set_copy_threads")

    def last_timings(self):
        arr = (C.c_double * 8)()
        self.lib.zkmi_last_timings(self.h, arr)
        return list(arr)
