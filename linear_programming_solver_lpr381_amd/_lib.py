"""ctypes loader for liblpx.so -- the C ABI declared in include/lpx.h.

There is no CPU fallback anywhere in this package: if the shared library is missing, or no
gfx950 device is visible when a compute entry point is called, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# LPX_LIB_PATH: diagnostic builds only (e.g. the -DLPX_STAMPS library of tools/diag_select_stamps.py)
LIB_PATH = os.environ.get("LPX_LIB_PATH") or os.path.join(_PKG, "liblpx.so")

# status / error codes (include/lpx.h)
OPTIMAL, UNBOUNDED, INFEASIBLE, ITER_LIMIT, RUNNING = 0, 1, 2, 3, 4
CUT_INTEGER, CUT_INCOMPLETE, CUT_ERROR, CUT_NOT_OPTIMAL = 0, 10, 11, 12
EINVAL, EDEVICE, ENOMEM = -1, -2, -3
E_GE_PRESENT, E_NEG_RHS, E_REVISED_PRECOND, E_SINGULAR, E_KNAP_SHAPE, E_UNKNOWN_ALGO, E_PARSE = (
    -10, -11, -12, -13, -14, -15, -16)

STATUS_NAMES = {OPTIMAL: "OPTIMAL", UNBOUNDED: "UNBOUNDED", INFEASIBLE: "INFEASIBLE",
                ITER_LIMIT: "ITER_LIMIT"}

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
PIVOT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.c_int)


class Stats(C.Structure):
    _fields_ = [("pivots", C.c_int64), ("launches", C.c_int64), ("loop_ms", C.c_double),
                ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("update_ms_sum", C.c_double),
                ("update_launches", C.c_int64), ("fdf_pivots", C.c_int64), ("cleanup_pivots", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RunOpts(C.Structure):
    _fields_ = [("eps", C.c_double), ("ratio_tol", C.c_double), ("max_iter", C.c_int),
                ("fdf_guard", C.c_int), ("cleanup", C.c_int), ("batch", C.c_int),
                ("use_graph", C.c_int), ("profile", C.c_int), ("resident", C.c_int)]


ALLREDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
TEXT_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint8), C.c_int, C.c_int)


# test seams (include/lpx_test.h): only the CPU test-suite sets these
TEST_NODE_LP = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, C.c_int, C.c_int, ip, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp,
                           C.POINTER(C.c_int64))
TEST_KNAP_RELAX = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, ip, ip, C.POINTER(C.c_int8), dp, dp, ip, dp)


class Problem(C.Structure):
    _fields_ = [("sense", C.c_int), ("n", C.c_int), ("m", C.c_int), ("c", dp), ("A", dp), ("rel", ip), ("b", dp)]


class SolveOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("batch", C.c_int), ("render_iterations", C.c_int),
                ("dual_flags", C.c_int), ("bnb_mode", C.c_int), ("bnb_search", C.c_int),
                ("concurrent_nodes", C.c_int), ("rank", C.c_int), ("world", C.c_int),
                ("max_nodes", C.c_int64), ("allreduce_max", ALLREDUCE_CB), ("allreduce_user", C.c_void_p),
                ("text_cb", TEXT_CB), ("text_user", C.c_void_p),
                ("bnb_dive", C.c_int)]


class TestSeams(C.Structure):
    """include/lpx_test.h: test-only stand-ins for the device loops (never installed by the package itself)."""
    _fields_ = [("node_lp", TEST_NODE_LP), ("knap_relax", TEST_KNAP_RELAX), ("user", C.c_void_p),
                ("fail_after_nodes", C.c_int64)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("has_solution", C.c_int), ("optimal_value", C.c_double),
                ("n", C.c_int), ("x", dp), ("R", C.c_int), ("C", C.c_int), ("T", dp), ("basis", ip),
                ("n_pivots", C.c_int), ("trace", ip), ("report", C.c_char_p), ("summary", C.c_char_p),
                ("lp_solves", C.c_int64), ("nodes", C.c_int64), ("n_log", C.c_int), ("node_log", ip),
                ("node_z", dp), ("aux", C.c_double * 4), ("stats", Stats), ("n_cuts", C.c_int), ("cuts", dp)]


class Parsed(C.Structure):
    _fields_ = [("sense", C.c_int), ("n", C.c_int), ("m", C.c_int), ("c", dp), ("A", dp), ("rel", ip),
                ("b", dp), ("ragged", C.c_int)]


class LpxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"liblpx error {code}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    """Loads liblpx.so (built in-tree by __graft_entry__.build()). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lpx_abi_version.restype = C.c_int
    L.lpx_device_count.restype = C.c_int
    L.lpx_init.argtypes = [C.c_int]
    L.lpx_last_error.argtypes = [C.c_char_p, C.c_int]
    L.lpx_device_name.argtypes = [C.c_char_p, C.c_int]
    u8p = C.POINTER(C.c_uint8)
    L.lpx_comm_unique_id.argtypes = [u8p]
    L.lpx_comm_init.argtypes = [C.c_int, C.c_int, u8p]
    L.lpx_comm_init_tcp.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.lpx_comm_allreduce_max.argtypes = [dp, C.c_int]
    L.lpx_comm_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), dp, C.POINTER(C.c_int)]
    L.lpx_comm_destroy.argtypes = []
    L.lpx_default_opts.argtypes = [C.POINTER(RunOpts), C.c_int]
    L.lpx_default_opts.restype = None
    L.lpx_tableau_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.lpx_tableau_destroy.argtypes = [vp]
    L.lpx_tableau_destroy.restype = None
    L.lpx_tableau_upload.argtypes = [vp, dp, ip]
    L.lpx_tableau_download.argtypes = [vp, dp, ip]
    L.lpx_tableau_snapshot.argtypes = [vp]
    L.lpx_tableau_restore.argtypes = [vp]
    L.lpx_tableau_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int)]
    L.lpx_tableau_trace.argtypes = [vp, ip, C.c_int, C.POINTER(C.c_int)]
    L.lpx_tableau_shape.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lpx_primal_run.argtypes = [vp, C.POINTER(RunOpts), PIVOT_CB, vp, C.POINTER(Stats)]
    L.lpx_dual_run.argtypes = [vp, C.POINTER(RunOpts), PIVOT_CB, vp, C.POINTER(Stats)]
    L.lpx_forced_pivots_run.argtypes = [vp, ip, ip, C.c_int, C.c_double, ip, C.POINTER(RunOpts),
                                        C.POINTER(Stats)]
    L.lpx_primal_tableau.argtypes = [dp, C.c_int, C.c_int, ip, C.c_double, C.c_int, PIVOT_CB, vp,
                                     C.POINTER(Stats)]
    L.lpx_dual_tableau.argtypes = [dp, C.c_int, C.c_int, ip, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, PIVOT_CB, vp, C.POINTER(Stats)]
    L.lpx_revised_create.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.POINTER(vp)]
    L.lpx_revised_destroy.argtypes = [vp]
    L.lpx_revised_destroy.restype = None
    L.lpx_revised_run.argtypes = [vp, C.POINTER(RunOpts), PIVOT_CB, vp, C.POINTER(Stats)]
    L.lpx_revised_result.argtypes = [vp, ip, ip, dp, dp]
    L.lpx_revised_binv.argtypes = [vp, dp]
    L.lpx_revised_iteration_view.argtypes = [vp, dp, dp]
    _sens = [C.POINTER(Problem), dp, C.c_int, C.c_int, ip]
    L.lpx_sensitivity_range_report.argtypes = _sens + [C.c_char_p, C.c_char_p, C.c_int]
    L.lpx_sensitivity_range.argtypes = _sens + [C.c_char_p, dp, dp]
    L.lpx_sensitivity_apply_change.argtypes = _sens + [C.c_char_p, C.c_double, ip, ip, C.c_char_p, C.c_int]
    L.lpx_sensitivity_shadow_prices.argtypes = _sens + [C.c_char_p, C.c_int]
    L.lpx_sensitivity_solve_duality.argtypes = _sens + [C.POINTER(SolveOpts), C.POINTER(Result)]
    L.lpx_revised_profile.argtypes = [vp, C.c_int, dp, C.POINTER(C.c_int)]
    L.lpx_revised_refactor.argtypes = [vp]
    L.lpx_revised_set_refactor_mode.argtypes = [vp, C.c_int]
    L.lpx_revised_set_drift_policy.argtypes = [vp, C.c_int, C.c_double]
    L.lpx_revised_residual.argtypes = [vp, dp, dp]
    L.lpx_revised_refactor_stats.argtypes = [vp, ip, ip, ip, dp, dp, ip]
    L.lpx_revised_set_refactor.argtypes = [vp, C.c_int]
    L.lpx_invert.argtypes = [dp, C.c_int, dp]
    L.lpx_revised_trace.argtypes = [vp, ip, C.c_int, C.POINTER(C.c_int)]
    L.lpx_revised_solve.argtypes = [dp, C.c_int, C.c_int, dp, dp, ip, ip, dp, dp, C.c_double, C.c_int,
                                    PIVOT_CB, vp, C.POINTER(Stats)]
    L.lpx_tableau_solution.argtypes = [vp, C.c_int, dp, dp]
    L.lpx_tableau_set_shape.argtypes = [vp, C.c_int, C.c_int]
    L.lpx_tableau_build_child.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double]
    L.lpx_tableau_basis.argtypes = [vp, ip]
    L.lpx_tableau_solution2.argtypes = [vp, C.c_int, dp, dp, ip]
    L.lpx_store_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.lpx_store_destroy.argtypes = [vp]
    L.lpx_store_destroy.restype = None
    L.lpx_store_save.argtypes = [vp, vp, C.POINTER(C.c_int)]
    L.lpx_store_release.argtypes = [vp, C.c_int]
    L.lpx_store_save_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    L.lpx_multi_solution.argtypes = [C.POINTER(vp), C.c_int, C.c_int, dp, dp, ip, C.c_int]
    L.lpx_tableau_build_child_from_store.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    L.lpx_tableau_build_node.argtypes = [vp, vp, C.c_int, ip, dp, dp, dp]
    L.lpx_tableau_build_children_from_store.argtypes = [C.POINTER(vp), C.POINTER(vp), ip, C.c_int, ip, ip, ip, dp]
    L.lpx_tableau_build_nodes.argtypes = [C.POINTER(vp), vp, C.c_int, ip, ip, dp, dp, dp]
    L.lpx_multi_run.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.POINTER(RunOpts), C.POINTER(RunOpts),
                                C.POINTER(C.c_int), C.POINTER(Stats)]
    L.lpx_multi_run_some.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.POINTER(RunOpts), C.POINTER(RunOpts),
                                     C.POINTER(C.c_int), C.POINTER(Stats), C.c_int]
    L.lpx_multi_run_begin.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.POINTER(RunOpts), C.POINTER(RunOpts), C.c_int]
    L.lpx_multi_run_end.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(Stats)]
    L.lpx_knapsack_create.argtypes = [dp, dp, C.c_int, C.c_double, C.POINTER(vp)]
    L.lpx_knapsack_destroy.argtypes = [vp]
    L.lpx_knapsack_destroy.restype = None
    L.lpx_knapsack_order.argtypes = [vp, ip]
    L.lpx_knapsack_relax_batch.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_int8), dp, dp, ip, dp]
    L.lpx_knapsack_relax_batch2.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_int8), dp, dp, ip, dp]
    L.lpx_test_set_seams.argtypes = [C.POINTER(TestSeams)]
    L.lpx_test_set_seams.restype = None
    L.lpx_test_comm_exchange_id.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_uint8)]
    L.lpx_knapsack_expand_batch.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), ip, C.POINTER(C.c_int8), C.POINTER(C.c_int64),
                                            dp, dp, ip, dp]
    L.lpx_knapsack_expand_begin.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), ip, C.POINTER(C.c_int8), C.POINTER(C.c_int64)]
    L.lpx_knapsack_expand_finish.argtypes = [vp, dp, dp, ip, dp]
    L.lpx_knapsack_node_list.argtypes = [vp, C.c_int64, ip, C.POINTER(C.c_int8), C.c_int, C.POINTER(C.c_int)]
    L.lpx_default_solve_opts.argtypes = [C.POINTER(SolveOpts)]
    L.lpx_default_solve_opts.restype = None
    L.lpx_solve.argtypes = [C.POINTER(Problem), C.c_char_p, C.POINTER(SolveOpts), C.POINTER(Result)]
    L.lpx_result_free.argtypes = [C.POINTER(Result)]
    L.lpx_result_free.restype = None
    L.lpx_parse_text.argtypes = [C.c_char_p, C.POINTER(Parsed)]
    L.lpx_parsed_free.argtypes = [C.POINTER(Parsed)]
    L.lpx_parsed_free.restype = None
    L.lpx_format_number.argtypes = [C.c_double, C.c_char_p, C.c_int]
    _lib = L
    return L


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().lpx_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc: int) -> int:
    """Raises LpxError for negative return codes, passes statuses through."""
    if rc < 0:
        raise LpxError(rc, last_error())
    return rc


def default_opts(dual: bool = False, **kw) -> RunOpts:
    o = RunOpts()
    lib().lpx_default_opts(C.byref(o), 1 if dual else 0)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown run option {k!r}")
        setattr(o, k, v)
    return o


NULL_CB = C.cast(None, PIVOT_CB)
