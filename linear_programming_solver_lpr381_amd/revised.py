"""Device-resident revised primal simplex: Python face of the lpx_revised_* C ABI (include/lpx.h)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import RunOpts, Stats, check, default_opts, dp, ip, lib
from .tableau import PivotCallback, _wrap_cb


class DeviceRevised:
    """[[B^-1, x_B], [c_B B^-1, z]] and A^T resident in HBM (Models/RevisedPrimalSimplex.cs:28-61).

    `A` is the m x n block of structural columns, `c` the costs of the standardised minimisation
    (c = -C for a Max model, :153-154), `b` the right-hand sides (>= 0).
    """

    def __init__(self, A: np.ndarray, c: np.ndarray, b: np.ndarray):
        A = np.ascontiguousarray(A, dtype=np.float64)
        c = np.ascontiguousarray(c, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        m, n = A.shape
        assert c.shape == (n,) and b.shape == (m,)
        self.m, self.n = m, n
        self._h = C.c_void_p()
        check(lib().lpx_revised_create(m, n, A.ctypes.data_as(dp), c.ctypes.data_as(dp), b.ctypes.data_as(dp),
                                       C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().lpx_revised_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def run(self, opts: Optional[RunOpts] = None, cb: Optional[PivotCallback] = None, **kw) -> Tuple[int, dict]:
        o = opts if opts is not None else default_opts(True, **kw)
        st = Stats()
        c = _wrap_cb(cb)
        rc = check(lib().lpx_revised_run(self._h, C.byref(o), c, None, C.byref(st)))
        return rc, st.as_dict()

    def result(self):
        """Returns (Bidx[m], Nidx[n] in list order, xB[m], z = c_B . x_B)."""
        Bidx = np.zeros(self.m, np.int32)
        Nidx = np.zeros(self.n, np.int32)
        xB = np.zeros(self.m, np.float64)
        z = C.c_double()
        check(lib().lpx_revised_result(self._h, Bidx.ctypes.data_as(ip), Nidx.ctypes.data_as(ip),
                                       xB.ctypes.data_as(dp), C.byref(z)))
        return Bidx, Nidx, xB, z.value

    def binv(self) -> np.ndarray:
        B = np.zeros((self.m, self.m), np.float64)
        check(lib().lpx_revised_binv(self._h, B.ctypes.data_as(dp)))
        return B

    def trace(self) -> np.ndarray:
        n = C.c_int()
        check(lib().lpx_revised_trace(self._h, None, 0, C.byref(n)))
        tr = np.zeros((max(n.value, 1), 2), np.int32)
        check(lib().lpx_revised_trace(self._h, tr.ctypes.data_as(ip), n.value, C.byref(n)))
        return tr[: n.value].copy()

    def refactor(self):
        """K7': recompute B^-1, x_B, pi, z from the current basis (device Gauss-Jordan, Invert :402-456)."""
        check(lib().lpx_revised_refactor(self._h))

    def set_refactor(self, every: int):
        check(lib().lpx_revised_set_refactor(self._h, int(every)))

    def set_refactor_mode(self, mode: int):
        """0 = exact (the reference's Invert, bit for bit), 1 = fast (Newton-Schulz on the FP64 matrix cores)."""
        check(lib().lpx_revised_set_refactor_mode(self._h, int(mode)))

    def set_drift_policy(self, check_every: int, tol: float = 1e-9):
        """Residual check of the maintained inverse every `check_every` iterations (0 = off); refactorise above `tol`."""
        check(lib().lpx_revised_set_drift_policy(self._h, int(check_every), float(tol)))

    def residual(self) -> Tuple[float, float]:
        """(rho, max_i |(B x_B)_i - b_i|) with rho = that maximum / (1 + max |b_i|), evaluated on the device."""
        rel, ab = C.c_double(), C.c_double()
        check(lib().lpx_revised_residual(self._h, C.byref(rel), C.byref(ab)))
        return rel.value, ab.value

    def profile(self, iters: int = 200) -> dict:
        """lpx_revised_profile: mean HIP-event microseconds per kernel of the four-launch iteration over `iters` real iterations."""
        us = (C.c_double * 4)()
        n = C.c_int()
        check(lib().lpx_revised_profile(self._h, iters, us, C.byref(n)))
        return {"rv_price": us[0], "rv_pick": us[1], "rv_upd_ftran": us[2], "rv_select2": us[3], "iterations": n.value}

    def refactor_stats(self) -> dict:
        a, b, c_, d, g, gc = C.c_int(), C.c_int(), C.c_int(), C.c_double(), C.c_double(), C.c_int()
        check(lib().lpx_revised_refactor_stats(self._h, C.byref(a), C.byref(b), C.byref(c_), C.byref(d), C.byref(g), C.byref(gc)))
        return {"refactors": a.value, "fast_steps": b.value, "fast_fallbacks": c_.value, "last_residual": d.value,
                "gemm_ms": g.value, "gemm_calls": gc.value}


def invert(M: np.ndarray) -> np.ndarray:
    """lpx_invert: the reference's Invert (Models/RevisedPrimalSimplex.cs:402-456) on the GPU, bit for bit."""
    M = np.ascontiguousarray(M, dtype=np.float64)
    assert M.ndim == 2 and M.shape[0] == M.shape[1]
    inv = np.zeros_like(M)
    check(lib().lpx_invert(M.ctypes.data_as(dp), M.shape[0], inv.ctypes.data_as(dp)))
    return inv
