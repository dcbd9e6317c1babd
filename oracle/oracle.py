"""ctypes binding of the CPU oracle (oracle/liblpx_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package.  See oracle/lpx_oracle.h for what each
entry point restates (reference file:line) and for the "parity unpinned" statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblpx_oracle.so")

OPTIMAL, UNBOUNDED, INFEASIBLE, ITER_LIMIT = 0, 1, 2, 3
E_GE_PRESENT, E_NEG_RHS, E_REVISED_PRECOND, E_SINGULAR, E_KNAP_SHAPE = -10, -11, -12, -13, -14
MAX, MIN = 0, 1
LE, GE, EQ = 0, 1, 2
DUAL_FAITHFUL, DUAL_REPAIRED = 0, 7

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build(force: bool = False) -> str:
    """Compile the oracle with its Makefile (gcc -O2 -ffp-contract=off)."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class _Problem(C.Structure):
    _fields_ = [("sense", C.c_int), ("n", C.c_int), ("m", C.c_int), ("c", _dp), ("A", _dp),
                ("rel", _ip), ("b", _dp)]


class _Result(C.Structure):
    _fields_ = [("status", C.c_int), ("has_solution", C.c_int), ("z", C.c_double), ("n", C.c_int),
                ("x", _dp), ("R", C.c_int), ("C", C.c_int), ("T", _dp), ("basis", _ip),
                ("n_pivots", C.c_int), ("trace", _ip), ("n_fdf_pivots", C.c_int)]


class _RevResult(C.Structure):
    _fields_ = [("status", C.c_int), ("n", C.c_int), ("m", C.c_int), ("z_original", C.c_double),
                ("z_internal", C.c_double), ("x", _dp), ("Bidx", _ip), ("Nidx", _ip), ("xB", _dp),
                ("n_iters", C.c_int), ("trace", _ip)]


class _BnbResult(C.Structure):
    _fields_ = [("status", C.c_int), ("best_z", C.c_double), ("n", C.c_int), ("best_x", _dp),
                ("has_incumbent", C.c_int), ("lp_solves", C.c_int64), ("nodes_visited", C.c_int64),
                ("total_pivots", C.c_int64), ("max_depth_seen", C.c_int), ("n_log", C.c_int),
                ("log_depth", _ip), ("log_outcome", _ip), ("log_branch_var", _ip), ("log_z", _dp)]


class _KnapResult(C.Structure):
    _fields_ = [("status", C.c_int), ("best_z", C.c_double), ("n", C.c_int), ("best_x", _ip),
                ("nodes_popped", C.c_int64), ("nodes_expanded", C.c_int64),
                ("relaxations", C.c_int64), ("max_heap", C.c_int64)]


class _CutResult(C.Structure):
    _fields_ = [("status", C.c_int), ("error", C.c_int), ("n", C.c_int), ("n_cuts", C.c_int), ("cut_A", _dp),
                ("cut_b", _dp), ("x", _dp), ("z", C.c_double), ("lp_solves", C.c_int64), ("total_pivots", C.c_int64)]


class _Parsed(C.Structure):
    _fields_ = [("sense", C.c_int), ("n", C.c_int), ("m", C.c_int), ("c", _dp), ("A", _dp),
                ("rel", _ip), ("b", _dp), ("ragged", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_choose_entering.argtypes = [_dp, C.c_int, C.c_int, C.c_double]
        L.orc_choose_leaving.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_pivot.argtypes = [_dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_pivot.restype = None
        L.orc_primal_tableau.argtypes = [_dp, C.c_int, C.c_int, _ip, C.c_double, C.c_int, _ip,
                                         C.POINTER(C.c_int)]
        L.orc_primal_tableau_mt.argtypes = [_dp, C.c_int, C.c_int, _ip, C.c_double, C.c_int, _ip,
                                            C.POINTER(C.c_int), C.c_int]
        L.orc_dual_tableau.argtypes = [_dp, C.c_int, C.c_int, _ip, C.c_double, C.c_double, C.c_int,
                                       C.c_int, C.c_int, _ip, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_forced_pivots.argtypes = [_dp, C.c_int, C.c_int, _ip, _ip, C.c_int, C.c_double, _ip]
        L.orc_forced_pivots.restype = None
        L.orc_primal_solve.argtypes = [C.POINTER(_Problem), C.c_int, C.POINTER(_Result)]
        L.orc_dual_solve.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int, C.POINTER(_Result)]
        L.orc_result_free.argtypes = [C.POINTER(_Result)]
        L.orc_result_free.restype = None
        L.orc_revised_solve.argtypes = [C.POINTER(_Problem), C.c_int, C.POINTER(_RevResult)]
        L.orc_revised_result_free.argtypes = [C.POINTER(_RevResult)]
        L.orc_revised_result_free.restype = None
        L.orc_invert.argtypes = [_dp, C.c_int, _dp]
        L.orc_bnb_solve.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int, C.c_int64, C.POINTER(_BnbResult)]
        L.orc_bnbr_solve.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int, C.c_int64, C.POINTER(_BnbResult)]
        L.orc_bnb_result_free.argtypes = [C.POINTER(_BnbResult)]
        L.orc_bnb_result_free.restype = None
        L.orc_knapsack_solve.argtypes = [C.POINTER(_Problem), C.c_int64, C.POINTER(_KnapResult)]
        L.orc_knap_result_free.argtypes = [C.POINTER(_KnapResult)]
        L.orc_knap_result_free.restype = None
        L.orc_knapsack_order.argtypes = [_dp, _dp, C.c_int, _ip]
        L.orc_knapsack_order.restype = None
        L.orc_knapsack_relax.argtypes = [_dp, _dp, C.c_int, C.c_double, _ip, _ip, _dp, _dp, _dp, _ip]
        L.orc_knapsack_relax.restype = None
        L.orc_cutting_plane.argtypes = [C.POINTER(_Problem), C.c_int, C.POINTER(_CutResult)]
        L.orc_cutting_plane_revised.argtypes = [C.POINTER(_Problem), C.c_int, C.POINTER(_CutResult)]
        L.orc_cut_result_free.argtypes = [C.POINTER(_CutResult)]
        L.orc_cut_result_free.restype = None
        L.orc_sens_range.argtypes = [C.POINTER(_Problem), _dp, C.c_int, C.c_int, _ip, C.c_int, C.c_int, _dp, _dp, _ip]
        L.orc_sens_shadow_prices.argtypes = [C.POINTER(_Problem), _dp, C.c_int, C.c_int, _dp]
        L.orc_sens_shadow_prices.restype = None
        L.orc_parse_text.argtypes = [C.c_char_p, C.POINTER(_Parsed), C.c_char_p, C.c_int]
        L.orc_parsed_free.argtypes = [C.POINTER(_Parsed)]
        L.orc_parsed_free.restype = None
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _cp(ptr, n, dtype):
    if not ptr or n <= 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


@dataclass
class Problem:
    """Dense LPProblem (Models/PrimalSimplex.cs:20-36)."""
    sense: int
    c: np.ndarray
    A: np.ndarray
    rel: np.ndarray
    b: np.ndarray

    def __post_init__(self):
        self.c = np.ascontiguousarray(self.c, dtype=np.float64)
        self.A = np.ascontiguousarray(self.A, dtype=np.float64).reshape(len(self.b), len(self.c))
        self.rel = np.ascontiguousarray(self.rel, dtype=np.int32)
        self.b = np.ascontiguousarray(self.b, dtype=np.float64)

    @property
    def n(self):
        return len(self.c)

    @property
    def m(self):
        return len(self.b)

    def _c(self):
        return _Problem(self.sense, self.n, self.m, _d(self.c), _d(self.A), _i(self.rel), _d(self.b))


@dataclass
class Result:
    status: int
    has_solution: bool = False
    z: float = 0.0
    x: Optional[np.ndarray] = None
    T: Optional[np.ndarray] = None
    basis: Optional[np.ndarray] = None
    trace: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.int32))
    n_fdf: int = 0


def _take_result(rc, r):
    L = lib()
    res = Result(status=rc, has_solution=bool(r.has_solution), z=r.z, n_fdf=r.n_fdf_pivots)
    if r.has_solution:
        res.x = _cp(r.x, r.n, np.float64)
        res.T = _cp(r.T, r.R * r.C, np.float64).reshape(r.R, r.C)
        res.basis = _cp(r.basis, r.R - 1, np.int32)
    res.trace = _cp(r.trace, 2 * r.n_pivots, np.int32).reshape(-1, 2) if r.trace else np.zeros((0, 2), np.int32)
    L.orc_result_free(C.byref(r))
    return res


def primal_solve(p: Problem, max_iter: int = 10000) -> Result:
    r = _Result()
    cp = p._c()
    rc = lib().orc_primal_solve(C.byref(cp), max_iter, C.byref(r))
    return _take_result(rc, r)


def dual_solve(p: Problem, flags: int = 0, max_iter: int = 10000) -> Result:
    r = _Result()
    cp = p._c()
    rc = lib().orc_dual_solve(C.byref(cp), flags, max_iter, C.byref(r))
    return _take_result(rc, r)


def choose_entering(T, eps=1e-9):
    T = np.ascontiguousarray(T, np.float64)
    return lib().orc_choose_entering(_d(T), T.shape[0], T.shape[1], eps)


def choose_leaving(T, q, eps=1e-9, tol=1e-9):
    T = np.ascontiguousarray(T, np.float64)
    return lib().orc_choose_leaving(_d(T), T.shape[0], T.shape[1], q, eps, tol)


def pivot(T, r, q):
    """In-place Gauss-Jordan pivot on a C-contiguous float64 tableau."""
    assert T.flags.c_contiguous and T.dtype == np.float64
    lib().orc_pivot(_d(T), T.shape[0], T.shape[1], r, q)


def primal_tableau(T, basis, eps=1e-9, max_iter=10000, threads=None):
    """Runs the primal loop in place. Returns (status, trace[k,2]).  threads=None: the scalar
    single-thread loop; threads=k (0 = all cores): OpenMP over the rows of Pivot, same bits."""
    assert T.flags.c_contiguous and T.dtype == np.float64 and basis.dtype == np.int32
    trace = np.zeros(2 * max(max_iter, 1), np.int32)
    n = C.c_int(0)
    if threads is None:
        st = lib().orc_primal_tableau(_d(T), T.shape[0], T.shape[1], _i(basis), eps, max_iter, _i(trace), C.byref(n))
    else:
        st = lib().orc_primal_tableau_mt(_d(T), T.shape[0], T.shape[1], _i(basis), eps, max_iter, _i(trace),
                                         C.byref(n), int(threads))
    return st, trace[: 2 * n.value].reshape(-1, 2).copy()


def dual_tableau(T, basis, eps=1e-9, ratio_tol=1e-12, fdf_guard=100, max_iter=10000, cleanup=0):
    assert T.flags.c_contiguous and T.dtype == np.float64 and basis.dtype == np.int32
    trace = np.zeros(2 * (max(fdf_guard, 0) + 2 * max(max_iter, 1) + 128), np.int32)   # FDF + dual loop + clean-up
    n = C.c_int(0)
    nf = C.c_int(0)
    st = lib().orc_dual_tableau(_d(T), T.shape[0], T.shape[1], _i(basis), eps, ratio_tol, fdf_guard,
                                max_iter, cleanup, _i(trace), C.byref(n), C.byref(nf))
    return st, trace[: 2 * n.value].reshape(-1, 2).copy(), nf.value


def forced_pivots(T, rows, cols, thresh=0.1):
    assert T.flags.c_contiguous and T.dtype == np.float64
    rows = np.ascontiguousarray(rows, np.int32)
    cols = np.ascontiguousarray(cols, np.int32)
    chosen = np.zeros(len(rows), np.int32)
    lib().orc_forced_pivots(_d(T), T.shape[0], T.shape[1], _i(rows), _i(cols), len(rows), thresh, _i(chosen))
    return chosen


@dataclass
class RevisedResult:
    status: int
    z_original: float
    z_internal: float
    x: np.ndarray
    Bidx: np.ndarray
    Nidx: np.ndarray
    xB: np.ndarray
    trace: np.ndarray


def revised_solve(p: Problem, max_iter: int = 10000) -> RevisedResult:
    r = _RevResult()
    cp = p._c()
    rc = lib().orc_revised_solve(C.byref(cp), max_iter, C.byref(r))
    if rc == E_REVISED_PRECOND:
        return RevisedResult(rc, 0.0, 0.0, np.zeros(0), np.zeros(0, np.int32), np.zeros(0, np.int32),
                             np.zeros(0), np.zeros((0, 2), np.int32))
    out = RevisedResult(rc, r.z_original, r.z_internal, _cp(r.x, r.n, np.float64),
                        _cp(r.Bidx, r.m, np.int32), _cp(r.Nidx, r.n, np.int32), _cp(r.xB, r.m, np.float64),
                        _cp(r.trace, 2 * r.n_iters, np.int32).reshape(-1, 2))
    lib().orc_revised_result_free(C.byref(r))
    return out


def invert(M):
    M = np.ascontiguousarray(M, np.float64)
    inv = np.zeros_like(M)
    rc = lib().orc_invert(_d(M), M.shape[0], _d(inv))
    return rc, inv


@dataclass
class BnbResult:
    status: int
    best_z: float
    best_x: np.ndarray
    has_incumbent: bool
    lp_solves: int
    nodes_visited: int
    total_pivots: int
    max_depth: int
    log: np.ndarray  # columns: depth, outcome, branch_var
    log_z: np.ndarray


def bnb_solve(p: Problem, mode: int = 0, max_iter: int = 10000, max_nodes: int = 0, revised: bool = False) -> BnbResult:
    r = _BnbResult()
    cp = p._c()
    (lib().orc_bnbr_solve if revised else lib().orc_bnb_solve)(C.byref(cp), mode, max_iter, max_nodes, C.byref(r))
    log = np.stack([_cp(r.log_depth, r.n_log, np.int32), _cp(r.log_outcome, r.n_log, np.int32),
                    _cp(r.log_branch_var, r.n_log, np.int32)], axis=1) if r.n_log else np.zeros((0, 3), np.int32)
    out = BnbResult(r.status, r.best_z, _cp(r.best_x, r.n, np.float64), bool(r.has_incumbent), r.lp_solves,
                    r.nodes_visited, r.total_pivots, r.max_depth_seen, log, _cp(r.log_z, r.n_log, np.float64))
    lib().orc_bnb_result_free(C.byref(r))
    return out


@dataclass
class KnapResult:
    rc: int
    status: int
    best_z: float
    best_x: np.ndarray
    nodes_popped: int
    nodes_expanded: int
    relaxations: int
    max_heap: int


def knapsack_solve(p: Problem, max_nodes: int = 0) -> KnapResult:
    r = _KnapResult()
    cp = p._c()
    rc = lib().orc_knapsack_solve(C.byref(cp), max_nodes, C.byref(r))
    if rc != 0:
        return KnapResult(rc, 1, float("-inf"), np.zeros(0, np.int32), 0, 0, 0, 0)
    out = KnapResult(rc, r.status, r.best_z, _cp(r.best_x, r.n, np.int32), r.nodes_popped, r.nodes_expanded,
                     r.relaxations, r.max_heap)
    lib().orc_knap_result_free(C.byref(r))
    return out


def knapsack_order(profit, weight):
    profit = np.ascontiguousarray(profit, np.float64)
    weight = np.ascontiguousarray(weight, np.float64)
    order = np.zeros(len(profit), np.int32)
    lib().orc_knapsack_order(_d(profit), _d(weight), len(profit), _i(order))
    return order


def knapsack_relax(profit, weight, cap, order, assigned, want_vector=False):
    profit = np.ascontiguousarray(profit, np.float64)
    weight = np.ascontiguousarray(weight, np.float64)
    order = np.ascontiguousarray(order, np.int32)
    assigned = np.ascontiguousarray(assigned, np.int32)
    n = len(profit)
    relaxed = np.zeros(n, np.float64) if want_vector else None
    p = C.c_double(0)
    w = C.c_double(0)
    f = C.c_int32(0)
    lib().orc_knapsack_relax(_d(profit), _d(weight), n, cap, _i(order), _i(assigned),
                             _d(relaxed) if want_vector else None, C.byref(p), C.byref(w), C.byref(f))
    return p.value, w.value, f.value, relaxed


def parse_text(text: str):
    """LPParser.ParseFromText. Returns (Problem, ragged) or raises ValueError(message)."""
    r = _Parsed()
    err = C.create_string_buffer(512)
    rc = lib().orc_parse_text(text.encode(), C.byref(r), err, 512)
    if rc != 0:
        raise ValueError(err.value.decode())
    p = Problem(r.sense, _cp(r.c, r.n, np.float64), _cp(r.A, r.m * r.n, np.float64).reshape(r.m, r.n),
                _cp(r.rel, r.m, np.int32), _cp(r.b, r.m, np.float64))
    ragged = bool(r.ragged)
    lib().orc_parsed_free(C.byref(r))
    return p, ragged


# ---- consumers of SimplexResult.Tableau / Basis (consumers.c) -----------------------------------------
CUT_INTEGER, CUT_INCOMPLETE, CUT_ERROR, CUT_NONBASIC, CUT_NOT_OPTIMAL = 0, 1, 2, 3, 4


@dataclass
class CutResult:
    status: int
    error: int
    cuts: np.ndarray        # [n_cuts, n+1] = (A, B) in the order added
    x: np.ndarray
    z: float
    lp_solves: int
    total_pivots: int


def cutting_plane(p: Problem, revised: bool = False, max_iter: int = 10000) -> CutResult:
    """CuttingPlane.Solve (Models/CuttingPlane.cs:13-139) / CuttingPlaneRevised.Solve (:14-78)."""
    r = _CutResult()
    cp = p._c()
    (lib().orc_cutting_plane_revised if revised else lib().orc_cutting_plane)(C.byref(cp), max_iter, C.byref(r))
    A = _cp(r.cut_A, r.n_cuts * r.n, np.float64).reshape(r.n_cuts, r.n)
    b = _cp(r.cut_b, r.n_cuts, np.float64).reshape(r.n_cuts, 1)
    out = CutResult(r.status, r.error, np.hstack([A, b]), _cp(r.x, r.n, np.float64), r.z, r.lp_solves, r.total_pivots)
    lib().orc_cut_result_free(C.byref(r))
    return out


def sens_range(p: Problem, T, basis, kind: int, index: int):
    """SensitivityAnalysis range numbers: kind 0 = constraint `index`, 1 = variable column `index`.
    Returns (rc, min, max, which) with which 0 constraint / 1 basic / 2 non-basic."""
    T = np.ascontiguousarray(T, np.float64)
    basis = np.ascontiguousarray(basis, np.int32)
    mn, mx, w = np.zeros(1), np.zeros(1), np.zeros(1, np.int32)
    cp = p._c()
    rc = lib().orc_sens_range(C.byref(cp), _d(T), T.shape[0], T.shape[1], _i(basis), kind, index, _d(mn), _d(mx), _i(w))
    return rc, float(mn[0]), float(mx[0]), int(w[0])


def sens_shadow_prices(p: Problem, T):
    T = np.ascontiguousarray(T, np.float64)
    out = np.zeros(len(p.b))
    cp = p._c()
    lib().orc_sens_shadow_prices(C.byref(cp), _d(T), T.shape[0], T.shape[1], _d(out))
    return out
