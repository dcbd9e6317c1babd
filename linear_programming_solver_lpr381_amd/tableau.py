"""Device-resident simplex tableau: thin Python face of the lpx_tableau_* C ABI (include/lpx.h)."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib
from ._lib import RunOpts, Stats, check, default_opts, dp, ip, lib

PivotCallback = Callable[[int, int, int], None]


def _wrap_cb(cb: Optional[PivotCallback]):
    if cb is None:
        return _lib.NULL_CB
    return _lib.PIVOT_CB(lambda _user, it, r, q: cb(it, r, q))


class DeviceTableau:
    """A simplex tableau living in HBM (row-major, padded leading dimension).

    Layout follows Models/PrimalSimplex.cs:179-203: R = m+1 rows, objective row last;
    C = n+m+1 columns, RHS last; basis[i] = column basic in row i.
    """

    def __init__(self, R: int, C_: int):
        self._h = C.c_void_p()
        check(lib().lpx_tableau_create(int(R), int(C_), C.byref(self._h)))
        self.R, self.C = int(R), int(C_)

    @classmethod
    def from_host(cls, T: np.ndarray, basis: Optional[np.ndarray] = None) -> "DeviceTableau":
        T = np.ascontiguousarray(T, dtype=np.float64)
        t = cls(T.shape[0], T.shape[1])
        t.upload(T, basis)
        return t

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().lpx_tableau_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def ld(self) -> int:
        ld = C.c_int()
        check(lib().lpx_tableau_shape(self._h, None, None, C.byref(ld)))
        return ld.value

    def device_ptr(self) -> Tuple[int, int]:
        p = C.c_void_p()
        ld = C.c_int()
        check(lib().lpx_tableau_device_ptr(self._h, C.byref(p), C.byref(ld)))
        return p.value, ld.value

    def upload(self, T: np.ndarray, basis: Optional[np.ndarray] = None):
        T = np.ascontiguousarray(T, dtype=np.float64)
        assert T.shape == (self.R, self.C), (T.shape, self.R, self.C)
        b = None
        if basis is not None:
            basis = np.ascontiguousarray(basis, dtype=np.int32)
            assert basis.shape == (self.R - 1,)
            b = basis.ctypes.data_as(ip)
        check(lib().lpx_tableau_upload(self._h, T.ctypes.data_as(dp), b))

    def download(self) -> Tuple[np.ndarray, np.ndarray]:
        T = np.empty((self.R, self.C), dtype=np.float64)
        basis = np.empty(max(self.R - 1, 0), dtype=np.int32)
        check(lib().lpx_tableau_download(self._h, T.ctypes.data_as(dp), basis.ctypes.data_as(ip)))
        return T, basis

    def snapshot(self):
        check(lib().lpx_tableau_snapshot(self._h))

    def restore(self):
        check(lib().lpx_tableau_restore(self._h))

    def trace(self) -> np.ndarray:
        n = C.c_int()
        check(lib().lpx_tableau_trace(self._h, None, 0, C.byref(n)))
        tr = np.zeros((max(n.value, 1), 2), dtype=np.int32)
        check(lib().lpx_tableau_trace(self._h, tr.ctypes.data_as(ip), n.value, C.byref(n)))
        return tr[: n.value].copy()

    def primal_run(self, opts: Optional[RunOpts] = None, cb: Optional[PivotCallback] = None,
                   **kw) -> Tuple[int, dict]:
        """PrimalSimplex loop (Models/PrimalSimplex.cs:92-124). Returns (status, stats)."""
        o = opts if opts is not None else default_opts(False, **kw)
        st = Stats()
        c = _wrap_cb(cb)
        rc = check(lib().lpx_primal_run(self._h, C.byref(o), c, None, C.byref(st)))
        return rc, st.as_dict()

    def dual_run(self, opts: Optional[RunOpts] = None, cb: Optional[PivotCallback] = None,
                 **kw) -> Tuple[int, dict]:
        """DualSimplex loop (Models/DualSimplex.cs:24,:36-113). Returns (status, stats)."""
        o = opts if opts is not None else default_opts(True, **kw)
        st = Stats()
        c = _wrap_cb(cb)
        rc = check(lib().lpx_dual_run(self._h, C.byref(o), c, None, C.byref(st)))
        return rc, st.as_dict()

    def forced_pivots(self, rows, cols, thresh: float = 0.1, opts: Optional[RunOpts] = None,
                      **kw) -> Tuple[np.ndarray, dict]:
        """Gauss-Jordan pivots (Models/PrimalSimplex.cs:245-257) at caller-chosen positions."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        assert rows.shape == cols.shape
        chosen = np.full(len(rows), -2, dtype=np.int32)
        o = opts if opts is not None else default_opts(False, **kw)
        st = Stats()
        check(lib().lpx_forced_pivots_run(self._h, rows.ctypes.data_as(ip), cols.ctypes.data_as(ip),
                                          len(rows), float(thresh), chosen.ctypes.data_as(ip),
                                          C.byref(o), C.byref(st)))
        return chosen, st.as_dict()


def primal_tableau(T: np.ndarray, basis: np.ndarray, eps: float = 1e-9, max_iter: int = 10000,
                   cb: Optional[PivotCallback] = None):
    """One-shot host-buffer entry point lpx_primal_tableau (in place). Returns (status, stats)."""
    assert T.flags.c_contiguous and T.dtype == np.float64 and basis.dtype == np.int32
    st = Stats()
    rc = check(lib().lpx_primal_tableau(T.ctypes.data_as(dp), T.shape[0], T.shape[1],
                                        basis.ctypes.data_as(ip), eps, max_iter, _wrap_cb(cb), None,
                                        C.byref(st)))
    return rc, st.as_dict()


def dual_tableau(T: np.ndarray, basis: np.ndarray, eps: float = 1e-9, ratio_tol: float = 1e-12,
                 fdf_guard: int = 100, max_iter: int = 10000, cleanup: int = 0,
                 cb: Optional[PivotCallback] = None):
    """One-shot host-buffer entry point lpx_dual_tableau (in place). Returns (status, stats)."""
    assert T.flags.c_contiguous and T.dtype == np.float64 and basis.dtype == np.int32
    st = Stats()
    rc = check(lib().lpx_dual_tableau(T.ctypes.data_as(dp), T.shape[0], T.shape[1],
                                      basis.ctypes.data_as(ip), eps, ratio_tol, fdf_guard, max_iter,
                                      cleanup, _wrap_cb(cb), None, C.byref(st)))
    return rc, st.as_dict()


def multi_run(tableaux, dual, primal_opts: Optional[RunOpts] = None, dual_opts: Optional[RunOpts] = None):
    """lpx_multi_run: runs several device tableaux to completion together (B&B node batches, K9).
    Returns (statuses, [stats dict])."""
    k = len(tableaux)
    hs = (C.c_void_p * k)(*[t._h for t in tableaux])
    dl = (C.c_int * k)(*[1 if d else 0 for d in dual])
    st = (C.c_int * k)()
    ss = (Stats * k)()
    po = primal_opts if primal_opts is not None else default_opts(False)
    do = dual_opts if dual_opts is not None else default_opts(True)
    check(lib().lpx_multi_run(hs, dl, k, C.byref(po), C.byref(do), st, ss))
    return list(st), [s.as_dict() for s in ss]
