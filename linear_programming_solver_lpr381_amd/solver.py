"""Python face of the reference's plugin boundary (Models/IPLAlgorithm.cs:5-8, Models/LPSolver.cs):
`LPSolver().Solve(problem, "Primal Simplex")`, `PrimalSimplex().Solve(problem)` ... with the same names
and error behaviour.  All solving goes through lpx_solve (C ABI) -> C++ host mirror -> HIP kernels.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from enum import IntEnum
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import lib


class Sense(IntEnum):       # Models/PrimalSimplex.cs:8
    Max = 0
    Min = 1


class Rel(IntEnum):         # Models/PrimalSimplex.cs:9
    LE = 0
    GE = 1
    EQ = 2


@dataclass
class Constraint:           # Models/PrimalSimplex.cs:11-18
    A: Sequence[float]
    Relation: Rel
    B: float


@dataclass
class LPProblem:            # Models/PrimalSimplex.cs:20-36
    ObjectiveSense: Sense = Sense.Max
    C: Sequence[float] = field(default_factory=list)
    Constraints: List[Constraint] = field(default_factory=list)

    @property
    def NumVars(self) -> int:
        return len(self.C)

    @classmethod
    def from_arrays(cls, sense, c, A, rel, b) -> "LPProblem":
        A = np.asarray(A, dtype=np.float64).reshape(len(b), len(c))
        return cls(Sense(int(sense)), list(map(float, c)),
                   [Constraint(A[i].tolist(), Rel(int(rel[i])), float(b[i])) for i in range(len(b))])


@dataclass
class SimplexResult:        # Models/PrimalSimplex.cs:38-49 (+ engine extras after VarNames)
    Report: str
    Summary: str
    OptimalValue: float
    Solution: Optional[np.ndarray]
    Tableau: Optional[np.ndarray]
    Basis: Optional[np.ndarray]
    VarNames: Optional[List[str]]
    Status: int = 0
    Trace: Optional[np.ndarray] = None
    LpSolves: int = 0
    Nodes: int = 0
    NodeLog: Optional[np.ndarray] = None
    NodeZ: Optional[np.ndarray] = None
    Aux: Optional[list] = None
    Stats: Optional[dict] = None
    Extra: Optional[np.ndarray] = None      # revised / knapsack: numbers the reference only prints
    Cuts: Optional[np.ndarray] = None       # cutting plane: rows (A[0..n), B) in the order added


class SolverException(Exception):
    """The reference's `throw new Exception(message)`; `.code` is the LPX_E_* of include/lpx.h."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def _problem_struct(p: LPProblem):
    n, m = p.NumVars, len(p.Constraints)
    c = np.ascontiguousarray(p.C, dtype=np.float64)
    A = np.zeros((max(m, 1), max(n, 1)), dtype=np.float64)
    rel = np.zeros(max(m, 1), dtype=np.int32)
    b = np.zeros(max(m, 1), dtype=np.float64)
    for i, k in enumerate(p.Constraints):
        if len(k.A) < n:
            # row.A[j] for j < n (Models/PrimalSimplex.cs:190)
            raise SolverException(_lib.EINVAL, "Index was outside the bounds of the array.")
        A[i, :n] = np.asarray(k.A[:n], dtype=np.float64)
        rel[i] = int(k.Relation)
        b[i] = float(k.B)
    st = _lib.Problem(int(p.ObjectiveSense), n, m, c.ctypes.data_as(_lib.dp), A.ctypes.data_as(_lib.dp),
                      rel.ctypes.data_as(_lib.ip), b.ctypes.data_as(_lib.dp))
    return st, (c, A, rel, b)


def _arr(ptr, n, dtype):
    if not ptr or n <= 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _solve_opts(engine: dict):
    """lpx_solve_opts from keyword engine options; returns (opts, objects to keep alive)."""
    L = lib()
    o = _lib.SolveOpts()
    L.lpx_default_solve_opts(C.byref(o))
    keep = []
    for k, v in engine.items():
        if k == "allreduce_max":
            if v is None:
                continue        # NULL: one process, or the library's own RCCL communicator (comm.py / lpx_comm_init)
            fn = v

            def _ar(_u, vals, count, fn=fn):
                a = np.ctypeslib.as_array(vals, shape=(count,))
                a[:] = fn(a.copy())
            cb = _lib.ALLREDUCE_CB(_ar)
            keep.append(cb)
            o.allreduce_max = cb
        elif k in ("test_node_lp", "test_knap_relax", "test_fail_after_nodes"):
            continue            # include/lpx_test.h seams: installed around the call by LPSolver.Solve
        elif hasattr(o, k):
            setattr(o, k, v)
        else:
            raise TypeError(f"unknown engine option {k!r}")
    return o, keep


def _take_result(r, n: int) -> "SimplexResult":
    """Copies an lpx_result into a SimplexResult and frees it. n = NumVars of the solved model."""
    try:
        has = bool(r.has_solution)
        T = _arr(r.T, r.R * r.C, np.float64).reshape(r.R, r.C) if r.R > 0 else None
        names = None
        if has and T is not None:
            ns = T.shape[1] - 1
            names = [f"x{j + 1}" for j in range(n)] + [f"c{j + 1}" for j in range(ns - n)]
        log = _arr(r.node_log, 3 * r.n_log, np.int32).reshape(-1, 3)
        res = SimplexResult(
            Report=(r.report or b"").decode(errors="replace"), Summary=(r.summary or b"").decode(errors="replace"),
            OptimalValue=r.optimal_value,
            Solution=_arr(r.x, r.n, np.float64) if has else None,
            Tableau=T if has else None,
            Basis=_arr(r.basis, max(r.R - 1, 0), np.int32) if has else None,
            VarNames=names, Status=r.status,
            Trace=_arr(r.trace, 2 * r.n_pivots, np.int32).reshape(-1, 2),
            LpSolves=r.lp_solves, Nodes=r.nodes, NodeLog=log, NodeZ=_arr(r.node_z, r.n_log, np.float64),
            Aux=list(r.aux), Stats=r.stats.as_dict(),
            Extra=None if has else (T.reshape(-1) if T is not None and T.size else _arr(r.x, r.n, np.float64)),
            Cuts=_arr(r.cuts, r.n_cuts * (n + 1), np.float64).reshape(-1, n + 1) if r.n_cuts > 0 else None)
        if not has and r.n > 0:
            res.Extra = _arr(r.x, r.n, np.float64)
        elif not has and T is not None:
            res.Extra = T.reshape(-1)
    finally:
        lib().lpx_result_free(C.byref(r))
    return res


class LPSolver:             # Models/LPSolver.cs:6-77
    def __init__(self, **engine):
        self.engine = engine
        self.FinalTableau = None

    def Solve(self, problem: LPProblem, algorithm: str,
              updatePivot: Optional[Callable[[str, Optional[np.ndarray]], None]] = None) -> SimplexResult:
        L = lib()
        o, keep = _solve_opts(self.engine)
        if updatePivot is not None:
            def _txt(_u, text, hl, R, Cc):
                mask = None
                if hl and R > 0:
                    mask = np.ctypeslib.as_array(hl, shape=(R * Cc,)).astype(bool).reshape(R, Cc)
                updatePivot(text.decode(errors="replace"), mask)
            tcb = _lib.TEXT_CB(_txt)
            keep.append(tcb)
            o.text_cb = tcb
        ps, hold = _problem_struct(problem)
        r = _lib.Result()
        seams = None
        if "test_node_lp" in self.engine or "test_knap_relax" in self.engine or "test_fail_after_nodes" in self.engine:      # test-only (include/lpx_test.h)
            seams = _lib.TestSeams()
            seams.fail_after_nodes = int(self.engine.get("test_fail_after_nodes", 0))
            if "test_node_lp" in self.engine:
                seams.node_lp = _lib.TEST_NODE_LP(self.engine["test_node_lp"])
            if "test_knap_relax" in self.engine:
                seams.knap_relax = _lib.TEST_KNAP_RELAX(self.engine["test_knap_relax"])
            L.lpx_test_set_seams(C.byref(seams))
        try:
            rc = L.lpx_solve(C.byref(ps), algorithm.encode() if algorithm is not None else b"", C.byref(o), C.byref(r))
        finally:
            if seams is not None:
                L.lpx_test_set_seams(None)
        if rc != 0:
            raise SolverException(rc, _lib.last_error())
        res = _take_result(r, problem.NumVars)
        self.FinalTableau = res.Tableau
        return res


class _Algo:                # ILPAlgorithm, Models/IPLAlgorithm.cs:5-8
    NAME = ""

    def __init__(self, **engine):
        self._solver = LPSolver(**engine)

    def Solve(self, problem: LPProblem, updatePivot=None) -> SimplexResult:
        return self._solver.Solve(problem, self.NAME, updatePivot)


class PrimalSimplex(_Algo):             # Models/PrimalSimplex.cs:52
    NAME = "Primal Simplex"


class RevisedPrimalSimplex(_Algo):      # Models/RevisedPrimalSimplex.cs:12
    NAME = "Revised Primal Simplex"


class DualSimplex(_Algo):               # Models/DualSimplex.cs:11
    NAME = "Dual Simplex"


class BranchAndBound(_Algo):            # Models/Branch&Bound.cs:20
    NAME = "Branch and Bound"


class BranchAndBoundRevised(_Algo):     # Models/BranchAndBoundRevised.cs:17
    NAME = "Revised Branch and Bound"


class BranchAndBoundKnapsack(_Algo):    # Models/BranchAndBoundKnapsack.cs:12
    NAME = "Branch and Bound Knapsack"


class CuttingPlane(_Algo):              # Models/CuttingPlane.cs:9 (built by Form1.cs:251, not by LPSolver)
    NAME = "Cutting Plane"


class CuttingPlaneRevised(_Algo):       # Models/CuttingPlaneRevised.cs:9 (Form1.cs:258)
    NAME = "Revised Cutting Plane"


class SensitivityAnalysis:
    """Models/SensitivityAnalysis.cs:11-297 over an LPProblem and the SimplexResult of solving it.
    The constructor checks of :24-43 are repeated by every call (they run in the library)."""

    def __init__(self, problem: LPProblem, result: SimplexResult, **engine):
        if problem is None:
            raise SolverException(_lib.EINVAL, "Value cannot be null. (Parameter 'problem')")
        if result is None:
            raise SolverException(_lib.EINVAL, "Value cannot be null. (Parameter 'result')")
        self.problem, self.result, self.engine = problem, result, engine
        self._call(lambda a: lib().lpx_sensitivity_shadow_prices(*a, None, 0))     # :24-43

    def _args(self):
        ps, hold = _problem_struct(self.problem)
        T = self.result.Tableau
        self._hold = hold
        if T is None:
            return [C.byref(ps), None, 0, 0, None], ps
        T = np.ascontiguousarray(T, dtype=np.float64)
        basis = np.ascontiguousarray(self.result.Basis if self.result.Basis is not None else [], dtype=np.int32)
        if len(basis) != T.shape[0] - 1:        # Basis length check of :40-41 needs the true length
            raise SolverException(_lib.EINVAL, f"Basis length invalid. Expected {T.shape[0] - 1}, got {len(basis)}.")
        self._hold = (hold, T, basis)
        return [C.byref(ps), T.ctypes.data_as(_lib.dp), T.shape[0], T.shape[1], basis.ctypes.data_as(_lib.ip)], ps

    def _call(self, fn):
        a, _ps = self._args()
        rc = fn(a)
        if rc < 0:
            raise SolverException(rc, _lib.last_error())
        return rc

    def _text(self, fn):
        n = self._call(lambda a: fn(a, None, 0))
        buf = C.create_string_buffer(n + 1)
        self._call(lambda a: fn(a, buf, n + 1))
        return buf.value.decode()

    def GetRangeReport(self, target: str) -> str:                       # :47-76
        t = target.encode()
        return self._text(lambda a, b, n: lib().lpx_sensitivity_range_report(*a, t, b, n))

    def GetRange(self, target: str):
        """The (min, max) GetRangeReport prints."""
        mn, mx = C.c_double(), C.c_double()
        self._call(lambda a: lib().lpx_sensitivity_range(*a, target.encode(), C.byref(mn), C.byref(mx)))
        return mn.value, mx.value

    def ApplyChange(self, target: str, value: float) -> str:            # :78-107 (mutates self.problem)
        t = target.encode()
        f, ix = C.c_int(-1), C.c_int(-1)
        msg = self._text(lambda a, b, n: lib().lpx_sensitivity_apply_change(*a, t, float(value), C.byref(f), C.byref(ix), b, n))
        if f.value == 0:
            self.problem.Constraints[ix.value].B = float(value)
        elif f.value == 1:
            c = list(self.problem.C)
            c[ix.value] = float(value)
            self.problem.C = c
        return msg

    def GetShadowPricesReport(self) -> str:                             # :109-128
        return self._text(lambda a, b, n: lib().lpx_sensitivity_shadow_prices(*a, b, n))

    def SolveUsingDuality(self) -> SimplexResult:                       # :130-219
        o, keep = _solve_opts(self.engine)
        r = _lib.Result()
        self._call(lambda a: lib().lpx_sensitivity_solve_duality(*a, C.byref(o), C.byref(r)))
        return _take_result(r, len(self.problem.Constraints))


def ParseFromText(text: str) -> LPProblem:
    """LPParser.ParseFromText (Models/LPParser.cs:9-79). Ragged rows are kept ragged."""
    L = lib()
    p = _lib.Parsed()
    rc = L.lpx_parse_text(text.encode(), C.byref(p))
    if rc != 0:
        raise SolverException(rc, _lib.last_error())
    try:
        c = _arr(p.c, p.n, np.float64)
        A = _arr(p.A, p.m * p.n, np.float64).reshape(p.m, p.n)
        rel = _arr(p.rel, p.m, np.int32)
        b = _arr(p.b, p.m, np.float64)
        prob = LPProblem.from_arrays(p.sense, c, A, rel, b)
        prob.ragged = bool(p.ragged)
    finally:
        L.lpx_parsed_free(C.byref(p))
    return prob


def format_number(v: float) -> str:
    buf = C.create_string_buffer(64)
    lib().lpx_format_number(float(v), buf, 64)
    return buf.value.decode()


class DeviceKnapsack:
    """Batched ComputeRelaxation (Models/BranchAndBoundKnapsack.cs:431-491) on the GPU."""

    def __init__(self, profit, weight, cap: float):
        self.profit = np.ascontiguousarray(profit, dtype=np.float64)
        self.weight = np.ascontiguousarray(weight, dtype=np.float64)
        self.n = len(self.profit)
        self._h = C.c_void_p()
        _lib.check(lib().lpx_knapsack_create(self.profit.ctypes.data_as(_lib.dp), self.weight.ctypes.data_as(_lib.dp),
                                             self.n, float(cap), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().lpx_knapsack_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def order(self) -> np.ndarray:
        o = np.zeros(self.n, np.int32)
        _lib.check(lib().lpx_knapsack_order(self._h, o.ctypes.data_as(_lib.ip)))
        return o

    def relax_batch2(self, nodes):
        """lpx_knapsack_relax_batch2: like relax_batch, and for every node also its two children (the node with its
        fractional item fixed to 0 / 1).  Returns arrays of shape (len(nodes), 3): column 0 the node, 1 and 2 the
        children; frac == -2 marks children that do not exist (no fractional item)."""
        off = [0]
        fidx, fval = [], []
        for nd in nodes:
            for i in sorted(nd):
                fidx.append(i)
                fval.append(nd[i])
            off.append(len(fidx))
        off = np.asarray(off, np.int32)
        fi = np.asarray(fidx if fidx else [0], np.int32)
        fv = np.asarray(fval if fval else [0], np.int8)
        k = len(nodes)
        p = np.zeros(3 * k); w = np.zeros(3 * k); fr = np.zeros(3 * k, np.int32); fx = np.zeros(3 * k)
        _lib.check(lib().lpx_knapsack_relax_batch2(self._h, k, off.ctypes.data_as(_lib.ip), fi.ctypes.data_as(_lib.ip),
                                                   fv.ctypes.data_as(C.POINTER(C.c_int8)), p.ctypes.data_as(_lib.dp),
                                                   w.ctypes.data_as(_lib.dp), fr.ctypes.data_as(_lib.ip),
                                                   fx.ctypes.data_as(_lib.dp)))
        return p.reshape(k, 3), w.reshape(k, 3), fr.reshape(k, 3), fx.reshape(k, 3)

    def expand_batch(self, parents, items, vals):
        """lpx_knapsack_expand_batch: node j = stored node parents[j] (-1 = root) + items[j] fixed to vals[j]; its list stays
        on the device.  Returns (ids, profit, weight, frac, fracval) with the last four of shape (len, 3) as relax_batch2;
        ids[j] + 1 / + 2 are the stored children of node j (fractional item fixed to 0 / 1)."""
        k = len(parents)
        par = np.asarray(parents, np.int64); it = np.asarray(items, np.int32); vv = np.asarray(vals, np.int8)
        ids = np.zeros(k, np.int64)
        p = np.zeros(3 * k); w = np.zeros(3 * k); fr = np.zeros(3 * k, np.int32); fx = np.zeros(3 * k)
        _lib.check(lib().lpx_knapsack_expand_batch(self._h, k, par.ctypes.data_as(C.POINTER(C.c_int64)), it.ctypes.data_as(_lib.ip),
                                                   vv.ctypes.data_as(C.POINTER(C.c_int8)), ids.ctypes.data_as(C.POINTER(C.c_int64)),
                                                   p.ctypes.data_as(_lib.dp), w.ctypes.data_as(_lib.dp), fr.ctypes.data_as(_lib.ip),
                                                   fx.ctypes.data_as(_lib.dp)))
        return ids, p.reshape(k, 3), w.reshape(k, 3), fr.reshape(k, 3), fx.reshape(k, 3)

    def node_list(self, node: int) -> dict:
        """The stored fixed list of a node as {item: 0/1}."""
        d = C.c_int()
        _lib.check(lib().lpx_knapsack_node_list(self._h, int(node), None, None, 0, C.byref(d)))
        idx = np.zeros(max(d.value, 1), np.int32); val = np.zeros(max(d.value, 1), np.int8)
        _lib.check(lib().lpx_knapsack_node_list(self._h, int(node), idx.ctypes.data_as(_lib.ip), val.ctypes.data_as(C.POINTER(C.c_int8)),
                                                d.value, C.byref(d)))
        assert list(idx[: d.value]) == sorted(idx[: d.value])
        return {int(i): int(v) for i, v in zip(idx[: d.value], val[: d.value])}

    def relax_batch(self, nodes):
        """nodes: list of dict {item_index: 0/1}. Returns (profit, weight, frac_sorted_idx, frac_value)."""
        off = [0]
        fidx, fval = [], []
        for nd in nodes:
            for i in sorted(nd):
                fidx.append(i)
                fval.append(nd[i])
            off.append(len(fidx))
        off = np.asarray(off, np.int32)
        fi = np.asarray(fidx if fidx else [0], np.int32)
        fv = np.asarray(fval if fval else [0], np.int8)
        k = len(nodes)
        p = np.zeros(k); w = np.zeros(k); fr = np.zeros(k, np.int32); fx = np.zeros(k)
        _lib.check(lib().lpx_knapsack_relax_batch(self._h, k, off.ctypes.data_as(_lib.ip), fi.ctypes.data_as(_lib.ip),
                                                  fv.ctypes.data_as(C.POINTER(C.c_int8)), p.ctypes.data_as(_lib.dp),
                                                  w.ctypes.data_as(_lib.dp), fr.ctypes.data_as(_lib.ip),
                                                  fx.ctypes.data_as(_lib.dp)))
        return p, w, fr, fx
