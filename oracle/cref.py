"""ctypes binding of the C oracle (oracle/c/libzkref.so) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_PATH = os.path.join(_DIR, "libzkref.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _DIR, "-s"])


class R1cs(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("n_wires", "n_public", "n_secret", "n_constraints",
                                          "n_coefs", "n_instr", "n_hints", "_pad")] + \
               [(n, C.c_void_p) for n in ("coefs", "l_ptr", "l_col", "l_cid", "r_ptr", "r_col",
                                          "r_cid", "o_ptr", "o_col", "o_cid", "instr",
                                          "solve_wire", "hint_kind", "hint_in_ptr", "hint_lc_ptr",
                                          "hint_col", "hint_cid", "hint_out_ptr", "hint_out")]


class Pk(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("log_n", "n_a", "n_b", "n_k", "n_z", "_pad")] + \
               [(n, C.c_void_p) for n in ("a_idx", "b_idx", "k_idx", "g1_a", "g1_b", "g1_k",
                                          "g1_z", "g2_b", "g1_alpha", "g1_beta", "g1_delta",
                                          "g2_beta", "g2_delta")]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a


def fr_mul(a, b):
    a, b = _u64(a), _u64(b)
    r = np.empty_like(a)
    lib().zkref_fr_mul(_p(a), _p(b), _p(r), C.c_size_t(a.size // 4))
    return r


def fq_mul(a, b):
    a, b = _u64(a), _u64(b)
    r = np.empty_like(a)
    lib().zkref_fq_mul(_p(a), _p(b), _p(r), C.c_size_t(a.size // 4))
    return r


def _un(name, a):
    a = _u64(a)
    r = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(r), C.c_size_t(a.size // 4))
    return r


def fr_to_mont(a):
    return _un("zkref_fr_to_mont", a)


def fr_from_mont(a):
    return _un("zkref_fr_from_mont", a)


def fq_to_mont(a):
    return _un("zkref_fq_to_mont", a)


def fq_from_mont(a):
    return _un("zkref_fq_from_mont", a)


def fr_inv(a):
    return _un("zkref_fr_inv", a)


def ntt(data, log_n, inverse=False, coset=False):
    """in place on a copy; data: [2^log_n, 4] uint64 Montgomery"""
    d = _u64(data).copy()
    lib().zkref_ntt(_p(d), log_n, int(inverse), int(coset))
    return d


def dft_naive(data, log_n, inverse=False, coset=False):
    d = _u64(data)
    out = np.empty_like(d)
    lib().zkref_dft_naive(_p(d), _p(out), log_n, int(inverse), int(coset))
    return out


def compute_h(a, b, c, log_n):
    a, b, c = _u64(a).copy(), _u64(b).copy(), _u64(c).copy()
    lib().zkref_compute_h(_p(a), _p(b), _p(c), log_n)
    return a


def msm(group, bases, scalars, c=0, naive=False):
    bases, scalars = _u64(bases), _u64(scalars)
    n = scalars.size // 4
    out = np.zeros(8 if group == 1 else 16, dtype=np.uint64)
    if naive:
        fn = lib().zkref_msm_g1_naive if group == 1 else lib().zkref_msm_g2_naive
        fn(_p(bases), _p(scalars), C.c_size_t(n), _p(out))
    else:
        fn = lib().zkref_msm_g1 if group == 1 else lib().zkref_msm_g2
        fn(_p(bases), _p(scalars), C.c_size_t(n), int(c), _p(out))
    return out


def batch_mul(group, base, scalars):
    base, scalars = _u64(base), _u64(scalars)
    n = scalars.size // 4
    out = np.zeros((n, 8 if group == 1 else 16), dtype=np.uint64)
    fn = lib().zkref_g1_batch_mul if group == 1 else lib().zkref_g2_batch_mul
    fn(_p(base), _p(scalars), C.c_size_t(n), _p(out))
    return out


def point_add(group, a, b):
    a, b = _u64(a), _u64(b)
    out = np.zeros_like(a)
    (lib().zkref_g1_add if group == 1 else lib().zkref_g2_add)(_p(a), _p(b), _p(out))
    return out
