"""ctypes wrapper of the plain-C oracle (oracle/clane_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Build with ``make -C oracle`` (``__graft_entry__.build()`` does it).  Only tests/, smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import torch

_DIR = Path(__file__).resolve().parent
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = _DIR / "libclane_oracle_c.so"
        if not so.exists():
            subprocess.run(["make", "-s", "-C", str(_DIR)], check=True)
        L = C.CDLL(str(so))
        p, i64, i32 = C.c_void_p, C.c_int64, C.c_int32
        L.clane_c_threads.restype = C.c_int
        L.clane_c_set_threads.argtypes = [C.c_int]
        L.clane_c_sweep_f32.restype = C.c_double
        L.clane_c_sweep_f32.argtypes = [p, p, p, i64, i32, p, p, C.c_float, p]
        L.clane_c_build_P_f32.restype = C.c_double
        L.clane_c_build_P_f32.argtypes = [p, p, i64, i32, p, p]
        L.clane_c_build_P_mode_f32.restype = C.c_double
        L.clane_c_build_P_mode_f32.argtypes = [p, p, i64, i32, p, p, C.c_int]
        _LIB = L
    return _LIB


def threads() -> int:
    return int(lib().clane_c_threads())


def set_threads(n: int) -> None:
    """OpenMP threads of the following calls (torchrun starts every rank with OMP_NUM_THREADS=1)."""
    lib().clane_c_set_threads(int(n))


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", torch.float32).contiguous()


def sweep(rowptr: np.ndarray, colidx: np.ndarray, P: torch.Tensor, X: torch.Tensor, Z: torch.Tensor, gamma: float,
          out: torch.Tensor = None):
    """(Z_new, sum|Z_new - Z|) -- embedder.py:84-94 in explicit loops, fp32."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    P, X, Z = _f32(P), _f32(X), _f32(Z)
    V, d = Z.shape
    Zn = out if out is not None else torch.empty_like(Z)
    delta = lib().clane_c_sweep_f32(rowptr.ctypes.data, colidx.ctypes.data, P.data_ptr(), V, d, X.data_ptr(),
                                    Z.data_ptr(), gamma, Zn.data_ptr())
    return Zn, float(delta)


def build_P(rowptr: np.ndarray, colidx: np.ndarray, Z: torch.Tensor, mode: str = "reference"):
    """(P values in CSR order, global denominator D) -- graph.py:118-128 + similarity.py:26-37, fp32.
    mode "per_edge" (the build's extension): true per-edge cosine scores; D is 0 then."""
    if mode not in ("reference", "per_edge"):
        raise ValueError(mode)
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    colidx = np.ascontiguousarray(colidx, dtype=np.int32)
    Z = _f32(Z)
    P = torch.empty(int(rowptr[-1]), dtype=torch.float32)
    D = lib().clane_c_build_P_mode_f32(rowptr.ctypes.data, colidx.ctypes.data, Z.shape[0], Z.shape[1], Z.data_ptr(),
                                       P.data_ptr(), int(mode == "per_edge"))
    return P, float(D)
