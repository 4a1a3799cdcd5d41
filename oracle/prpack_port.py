"""ctypes wrapper + build recipe for oracle/prpack_port.c.  TEST INFRASTRUCTURE ONLY."""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "prpack_port.c")
_LIB = os.path.join(_HERE, "libhrag_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc -O2 -shared -fPIC oracle/prpack_port.c -> oracle/libhrag_oracle.so"""
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-shared", "-fPIC", "-o", _LIB, _SRC, "-lm"])
    return _LIB


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB)
        i64p = ctypes.POINTER(ctypes.c_int64)
        i32p = ctypes.POINTER(ctypes.c_int32)
        f64p = ctypes.POINTER(ctypes.c_double)
        intp = ctypes.POINTER(ctypes.c_int)
        lib.hro_ppr_gs.argtypes = [ctypes.c_int64, i64p, i32p, f64p, f64p, ctypes.c_double,
                                   ctypes.c_double, ctypes.c_int, f64p, intp]
        lib.hro_ppr_gs.restype = ctypes.c_int
        lib.hro_ppr_ge.argtypes = [ctypes.c_int64, i64p, i32p, f64p, f64p, ctypes.c_double, f64p]
        lib.hro_ppr_ge.restype = ctypes.c_int
        lib.hro_ppr_prpack.argtypes = [ctypes.c_int64, i64p, i32p, f64p, f64p, ctypes.c_double,
                                       f64p, intp]
        lib.hro_ppr_prpack.restype = ctypes.c_int
        _lib = lib
    return _lib


class PrpackCSR:
    """Holds the arrays in the C layout so repeated solves do not re-convert."""

    def __init__(self, p: sp.csr_matrix):
        p = p.tocsr()
        self.n = p.shape[0]
        self.rowptr = np.ascontiguousarray(p.indptr, dtype=np.int64)
        self.col = np.ascontiguousarray(p.indices, dtype=np.int32)
        self.val = np.ascontiguousarray(p.data, dtype=np.float64)

    def _ptrs(self):
        return (self.rowptr.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)),
                self.col.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                self.val.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))

    def solve(self, reset, alpha: float = 0.5, method: str = "prpack"):
        """Returns (x, sweeps).  method: "prpack" (igraph dispatch), "gs", "ge"."""
        lib = _load()
        r = np.ascontiguousarray(reset, dtype=np.float64)
        x = np.empty(self.n, dtype=np.float64)
        sweeps = ctypes.c_int(0)
        rp, ci, va = self._ptrs()
        f64p = ctypes.POINTER(ctypes.c_double)
        if method == "prpack":
            rc = lib.hro_ppr_prpack(self.n, rp, ci, va, r.ctypes.data_as(f64p), alpha,
                                    x.ctypes.data_as(f64p), ctypes.byref(sweeps))
        elif method == "gs":
            rc = lib.hro_ppr_gs(self.n, rp, ci, va, r.ctypes.data_as(f64p), alpha, 1e-10, 1000,
                                x.ctypes.data_as(f64p), ctypes.byref(sweeps))
        elif method == "ge":
            rc = lib.hro_ppr_ge(self.n, rp, ci, va, r.ctypes.data_as(f64p), alpha,
                                x.ctypes.data_as(f64p))
        else:
            raise ValueError(method)
        if rc == 3:
            raise ValueError("reset vector has no positive entry")
        if rc != 0:
            raise RuntimeError(f"prpack_port failed with status {rc}")
        return x, sweeps.value
