/*
 * lpx.h -- C ABI of liblpx.so, the MI355X (gfx950) simplex / branch-and-bound engine.
 *
 * This is the drop-in boundary for the hot path of Jellyman750/Linear_Programming_Solver_LPR381.
 * The reference is managed C# with no native interface of its own; each entry point below names
 * the reference loop it replaces (paths relative to Linear_Programming_Solver/ in the reference)
 * and is what a P/Invoke shim inside the reference's ILPAlgorithm implementations
 * (Models/IPLAlgorithm.cs:5-8) would bind -- see INTEGRATION.md for that shim.
 *
 * Conventions
 *   - plain C types only; every buffer is caller-owned unless a *_free function is named;
 *   - tableaux are row-major `double[R*C]` exactly as C#'s `double[R,C]` (zero-copy under
 *     `fixed (double* p = T)`): R = m+1 rows with the objective row LAST, C = n+m+1 columns
 *     with the RHS column LAST (Models/PrimalSimplex.cs:179-203);
 *   - return value >= 0 is a solver status, < 0 an error; lpx_last_error() has the message;
 *   - nothing here falls back to the CPU: without a gfx950 device every compute entry point
 *     returns LPX_EDEVICE.
 */
#ifndef LPX_H
#define LPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPX_ABI_VERSION 1

/* ---- status (soft outcomes the reference reports as text) and errors (its exceptions) ------- */
enum {
    LPX_OPTIMAL    = 0,   /* "OPTIMAL"    Models/PrimalSimplex.cs:126 */
    LPX_UNBOUNDED  = 1,   /* "UNBOUNDED"  Models/PrimalSimplex.cs:102-106 */
    LPX_INFEASIBLE = 2,   /* "INFEASIBLE" Models/DualSimplex.cs:92-96 */
    LPX_ITER_LIMIT = 3,   /* exception "Iteration limit exceeded." Models/PrimalSimplex.cs:95-96 */
    LPX_RUNNING    = 4,   /* internal: loop not finished */
    /* outcomes of the cutting-plane consumers (lpx_result.status for "Cutting Plane" / "Revised Cutting Plane") */
    LPX_CUT_INTEGER     = 0,  /* "Status: OPTIMAL INTEGER" Models/CuttingPlane.cs:91-104, CuttingPlaneRevised.cs:49-57 */
    LPX_CUT_INCOMPLETE  = 10, /* 50 iterations used up, Models/CuttingPlane.cs:132-137, CuttingPlaneRevised.cs:70-77 */
    LPX_CUT_ERROR       = 11, /* "Error: ..." summaries, Models/CuttingPlane.cs:42-74,116-124 */
    LPX_CUT_NOT_OPTIMAL = 12, /* Models/CuttingPlaneRevised.cs:27-35 */
    LPX_EINVAL     = -1,
    LPX_EDEVICE    = -2,  /* no usable gfx950 device / HIP failure */
    LPX_ENOMEM     = -3,
    LPX_E_GE_PRESENT      = -10, /* Models/PrimalSimplex.cs:70 */
    LPX_E_NEG_RHS         = -11, /* Models/PrimalSimplex.cs:75 */
    LPX_E_REVISED_PRECOND = -12, /* Models/RevisedPrimalSimplex.cs:21 */
    LPX_E_SINGULAR        = -13, /* Models/RevisedPrimalSimplex.cs:426 */
    LPX_E_KNAP_SHAPE      = -14, /* Models/BranchAndBoundKnapsack.cs:66-69 */
    LPX_E_UNKNOWN_ALGO    = -15, /* Models/LPSolver.cs:39-42 */
    LPX_E_PARSE           = -16  /* Models/LPParser.cs exceptions */
};

/* Per-pivot event, fired on the calling thread between batches, in pivot order.  Replaces the
 * reference's per-iteration `updatePivot(text, bool[,])` callback (Models/PrimalSimplex.cs:113-121),
 * which formats the whole tableau; hosts that want to render it call lpx_tableau_download. */
typedef void (*lpx_pivot_cb)(void* user, int iter, int row, int col);

typedef struct lpx_stats {
    int64_t pivots;          /* completed pivots */
    int64_t launches;        /* kernel launches enqueued (including early-exit ones) */
    double  loop_ms;         /* host wall time of the device-resident loop (no H2D/D2H) */
    double  h2d_ms;          /* upload time of one-shot entry points */
    double  d2h_ms;
    double  update_ms_sum;   /* profile mode only: sum of HIP-event durations of the kernel that streams the tableau (the rank-1
                              * update; the fused update + select launch of lpx_primal_run) */
    int64_t update_launches; /* profile mode only: launches included in update_ms_sum */
    int64_t fdf_pivots;      /* dual: pivots spent in ForceDualFeasibility */
    int64_t cleanup_pivots;  /* dual, repaired mode: pivots of the primal clean-up phase */
} lpx_stats;

typedef struct lpx_run_opts {
    double eps;        /* Eps, 1e-9 (Models/PrimalSimplex.cs:55, Models/DualSimplex.cs:13) */
    double ratio_tol;  /* hysteresis of the ratio scans: 1e-9 primal (:235), 1e-12 dual (:85,:220) */
    int    max_iter;   /* 10000 (Models/PrimalSimplex.cs:54, Models/DualSimplex.cs:39) */
    int    fdf_guard;  /* dual: ForceDualFeasibility guard, 100 (Models/DualSimplex.cs:202) */
    int    cleanup;    /* dual: 1 = repaired-mode primal clean-up phase */
    int    batch;      /* pivots enqueued between host polls of the device state; 0 = default */
    int    use_graph;  /* 1 = replay a captured hipGraph per batch, 0 = eager launches */
    int    profile;    /* 1 = eager, every update kernel bracketed by HIP events (roofline leg) */
    int    resident;   /* primal loop with the whole tableau resident in LDS (one persistent workgroup per CU,
                          csrc/lpx_resident.hip) when it fits the chip: 0 = automatic, 1 = require it
                          (LPX_EINVAL when the tableau does not fit), -1 = never (streaming kernels) */
} lpx_run_opts;

void lpx_default_opts(lpx_run_opts* o, int dual);

/* ---- library / device ----------------------------------------------------------------------- */
int         lpx_abi_version(void);
int         lpx_device_count(void);              /* 0 when no GPU is visible */
int         lpx_init(int device);                /* binds this process to one GPU (one process per GPU) */
int         lpx_last_error(char* buf, int len);  /* copies the last error message of this thread */
int         lpx_device_name(char* buf, int len);

/* ---- X1: the incumbent exchange between the processes of a sharded search (one process per GPU) ------------------
 * Replaces, for a node queue sharded over the GPUs of one node, the compare-and-update of the single incumbent field the
 * reference keeps (`BestObjective`, Models/Branch&Bound.cs:182,191; `_bestValue`, Models/BranchAndBoundKnapsack.cs:124,
 * 157-160): ONE all-reduce(MAX) over FP64 per level / per round, RCCL over xGMI (ncclAllReduce, ncclMax, ncclDouble) on a
 * communicator the library owns.  RCCL is bound at run time (dlopen of librccl.so.1; LPX_RCCL_LIB overrides the path), so
 * single-GPU hosts need no RCCL.  Call after lpx_init(local GPU).  While a communicator exists, lpx_solve uses it for every
 * sharded search whose lpx_solve_opts.allreduce_max is NULL (opts.rank / opts.world must then equal the communicator's). */
#define LPX_COMM_ID_BYTES 128
int lpx_comm_unique_id(uint8_t* id /* [LPX_COMM_ID_BYTES] */);   /* rank 0; the host ships the bytes to the other ranks */
int lpx_comm_init(int rank, int world, const uint8_t* id);       /* collective over all ranks: ncclCommInitRank */
/* The same for hosts without a side channel: rank 0 serves the id on host:port (TCP), the others fetch it there. */
int lpx_comm_init_tcp(int rank, int world, const char* host, int port);
int lpx_comm_allreduce_max(double* vals, int count);             /* MAX over ranks, in place (host buffer) */
/* rank = -1 / world = 0 when no communicator exists; counters since lpx_comm_init.  Any pointer may be NULL. */
int lpx_comm_info(int* rank, int* world, int64_t* allreduces, double* allreduce_ms, int* rccl_version);
int lpx_comm_destroy(void);

/* ---- device-resident tableau ---------------------------------------------------------------- */
/* HBM layout: row-major with the leading dimension padded to a multiple of 16 doubles (128 B) so
 * that every row starts on a cache line and 16-byte vector accesses are aligned even for odd C. */
typedef struct lpx_tableau lpx_tableau;

int  lpx_tableau_create(int R, int C, lpx_tableau** out);
void lpx_tableau_destroy(lpx_tableau* t);
int  lpx_tableau_upload(lpx_tableau* t, const double* T, const int32_t* basis /* [R-1] or NULL */);
int  lpx_tableau_download(lpx_tableau* t, double* T, int32_t* basis /* may be NULL */);
int  lpx_tableau_snapshot(lpx_tableau* t);       /* keep a device copy of the current tableau+basis */
int  lpx_tableau_restore(lpx_tableau* t);        /* D2D restore from the snapshot, resets the loop state */
int  lpx_tableau_device_ptr(lpx_tableau* t, void** dptr, int* ld);
int  lpx_tableau_trace(lpx_tableau* t, int32_t* trace /* [2*cap] */, int cap, int* n);
int  lpx_tableau_shape(const lpx_tableau* t, int* R, int* C, int* ld);
/* A handle created for (Rcap, Ccap) can hold any smaller tableau: the kernels read the live shape from a
 * device record, so one handle (and its captured hipGraph) serves every depth of a B&B tree. */
int  lpx_tableau_set_shape(lpx_tableau* t, int R, int C);

/* The hot loops.  Each iteration is two launches on one stream:
 *   select  -- ChooseEntering + ChooseLeaving (+ pivot-row normalisation and pivot-column snapshot)
 *   update  -- the rank-1 Gauss-Jordan update T[i,:] -= T[i,q] * T[r,:]  (i != r)
 * or, for lpx_primal_run without a per-pivot callback, ONE: the update written out of place into a second tableau buffer
 * the library keeps beside the handle's own, with the next pivot's select in the same grid (the result is brought back
 * into the handle's buffer before the call returns; LPX_FUSED_PIVOT=0 in the environment keeps the two-launch form).  */

/* PrimalSimplex.Solve's while(true), Models/PrimalSimplex.cs:92-124
 * (ChooseEntering :205-220, ChooseLeaving :222-243, Pivot :245-257). */
int lpx_primal_run(lpx_tableau* t, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st);

/* DualSimplex.Solve steps 3 and 5, Models/DualSimplex.cs:24 + :36-113
 * (ForceDualFeasibility :195-228, leaving row :45-55, entering column :76-91, Pivot :232-246). */
int lpx_dual_run(lpx_tableau* t, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st);

/* Pivot (Models/PrimalSimplex.cs:245-257) on caller-chosen positions: for k in [0,count):
 * r = rows[k]; q = first column >= cols[k] (wrapping) with |T[r,q]| >= thresh; pivot(r,q).
 * chosen[k] = q or -1.  The headline rank-1-update benchmark and the bitwise kernel parity tests. */
int lpx_forced_pivots_run(lpx_tableau* t, const int32_t* rows, const int32_t* cols, int count,
                          double thresh, int32_t* chosen, const lpx_run_opts* o, lpx_stats* st);

/* x[basis[i]] = T[i,last] for basis[i] < nvars, *z = T[m,last] (FinalizeReport,
 * Models/PrimalSimplex.cs:130-138) without downloading the tableau. */
int lpx_tableau_solution(lpx_tableau* t, int nvars, double* x, double* z);

/* Node assembly on the device: `node` becomes the tableau
 * BuildTableau (Models/PrimalSimplex.cs:179-203) would produce for the root model plus `ncuts` unit
 * rows (Models/Branch&Bound.cs:233-248); the node handle needs capacity for root shape + ncuts and takes
 * that shape.  Row k has coef[k] at column var[k], zero[k] (a signed zero: the
 * reference's `A[j] *= -1` turns 0 into -0) elsewhere, its own slack, and rhs[k].  `root` holds the
 * prepared root tableau (its snapshot if one was taken) and is not modified.  Stream-ordered with
 * the run that follows on `node`. */
int lpx_tableau_build_node(lpx_tableau* node, const lpx_tableau* root, int ncuts, const int32_t* var,
                           const double* coef, const double* zero, const double* rhs);

/* The same for a group of nodes in one launch: node i gets the branching rows [cut_off[i], cut_off[i+1]) of the flattened
 * arrays (cut_off has count + 1 entries, cut_off[0] = 0).  Returns when the nodes are built. */
int lpx_tableau_build_nodes(lpx_tableau** nodes, const lpx_tableau* root, int count, const int32_t* cut_off,
                            const int32_t* var, const double* coef, const double* zero, const double* rhs);

/* Warm start of a branch-and-bound child (NOT what the reference does -- it re-solves every
 * node from the slack basis, Models/Branch&Bound.cs:148): `child` becomes `parent`'s final tableau plus the row
 * of `x_var <= bound` (is_ge = 0) or `x_var >= bound` (is_ge = 1) written in the parent's basis, where
 * `row_of_var` is the row in which x_var is basic.  The result is dual feasible; run it with lpx_dual_run /
 * lpx_multi_run and fdf_guard = 0.  `child` needs capacity for the parent's shape + 1. */
int lpx_tableau_build_child(lpx_tableau* child, lpx_tableau* parent, int var, int row_of_var, int is_ge, double bound);
int lpx_tableau_basis(lpx_tableau* t, int32_t* basis /* [R-1] */);
int lpx_tableau_solution2(lpx_tableau* t, int nvars, double* x, double* z, int32_t* basis_out /* [R-1] or NULL */);
/* The same for a batch of solved nodes in ONE launch + one wait (the reads of FinalizeReport, Models/PrimalSimplex.cs:135-138,
 * for every node of a branch-and-bound group): x is count x nvars, z has count entries, basis_out (or NULL) count x basis_stride. */
int lpx_multi_solution(lpx_tableau** ts, int count, int nvars, double* x, double* z, int32_t* basis_out, int basis_stride);
/* Parent store: a solved node parks its final tableau in a slab slot (one D2D copy) and gives its handle
 * back; its children are assembled from the slot.  Slots are sized for one capacity class (same Rcap/Ccap as
 * the handles that use the store) and allocated 128 at a time. */
typedef struct lpx_store lpx_store;
int  lpx_store_create(int Rcap, int Ccap, lpx_store** out);
void lpx_store_destroy(lpx_store* s);   /* its device chunks stay with the process for the next store (up to LPX_STORE_CACHE_GB, default 64) */
int  lpx_store_save(lpx_store* s, lpx_tableau* t, int* slot_out);
int  lpx_store_release(lpx_store* s, int slot);
/* The same for a batch of solved nodes (stores[i] / ts[i] / slots[i]): all copies are enqueued, then one wait per stream. */
int  lpx_store_save_multi(lpx_store** stores, lpx_tableau** ts, int count, int* slots);
int  lpx_tableau_build_child_from_store(lpx_tableau* child, lpx_store* s, int slot, int var, int row_of_var,
                                        int is_ge, double bound);

/* The same for a group of children in one launch (child i from stores[i] / slots[i]); returns when they are built. */
int  lpx_tableau_build_children_from_store(lpx_tableau** children, lpx_store** stores, const int* slots, int count,
                                           const int32_t* var, const int32_t* row_of_var, const int32_t* is_ge, const double* bound);

/* Branch-and-bound node batches: runs `count` independent tableaux to completion,
 * interleaving their batches on their own streams so that small node LPs overlap on one GPU.
 * dual[i] selects lpx_dual_run (1) or lpx_primal_run (0) semantics; statuses[i] gets each status. */
int lpx_multi_run(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* primal_opts,
                  const lpx_run_opts* dual_opts, int* statuses, lpx_stats* stats /* [count] or NULL */);

/* The same for a rolling batch: returns as soon as at most `min_active` runs are unfinished and reports those as LPX_RUNNING;
 * handed in again (with fresh tableaux beside them) they continue where they stopped.  Batched streaming kernels only. */
int lpx_multi_run_some(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* primal_opts,
                       const lpx_run_opts* dual_opts, int* statuses, lpx_stats* stats /* [count] or NULL */, int min_active);

/* The same in two halves, for hosts that keep TWO OR MORE rolling batches (slots 0 .. LPX_ASYNC_SLOTS - 1) so that one pivots while
 * another is read back, decided on and refilled: _begin enqueues `steps` pivots (rounded up to even) of every run of the batch -- fresh
 * tableaux and runs a previous window left as LPX_RUNNING alike -- on the slot's own stream and returns at once; _end waits for
 * that window and reports as lpx_multi_run_some does (LPX_RUNNING = unfinished, hand it in again).  Between the two calls the
 * batch's handles must not be touched.  _begin returns 1 (nothing enqueued) when the one-launch-per-step kernels cannot take
 * the batch (a second tableau buffer did not fit, profile mode): use lpx_multi_run_some then. */
#define LPX_ASYNC_SLOTS 4   /* slot = 0 .. LPX_ASYNC_SLOTS - 1 */
int lpx_multi_run_begin(int slot, lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* primal_opts,
                        const lpx_run_opts* dual_opts, int steps);
int lpx_multi_run_end(int slot, int* statuses, lpx_stats* stats /* [count of the _begin] or NULL */);

/* ---- one-shot entry points on host buffers (what the C# shim binds) -------------------------- */
/* Replaces the loop of PrimalSimplex.Solve (Models/PrimalSimplex.cs:92-124) on the `double[,]`
 * built by BuildTableau (:179-203).  T and basis are updated in place. */
int lpx_primal_tableau(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                       lpx_pivot_cb cb, void* user, lpx_stats* st);
/* Replaces ForceDualFeasibility + the loop of DualSimplex.Solve (Models/DualSimplex.cs:24,:36-113). */
int lpx_dual_tableau(double* T, int R, int C, int32_t* basis, double eps, double ratio_tol,
                     int fdf_guard, int max_iter, int cleanup,
                     lpx_pivot_cb cb, void* user, lpx_stats* st);

/* ---- revised primal simplex (device-resident) ------------------------------------------------ */
/* Replaces the loop of RevisedPrimalSimplex.Solve (Models/RevisedPrimalSimplex.cs:58-142):
 * pricing  r_N = c_N - (c_B B^-1) N  (MultiplyRow :365-378, Subtract :388-393),
 * entering = first strict minimum below -Eps in Nidx LIST order (:76-83),
 * direction d = B^-1 a_q (Multiply :325-336), ratio test with 1e-12 hysteresis (:99-112),
 * basis bookkeeping Bidx[r]=q; Nidx.RemoveAt(pos); Nidx.Add(leaving) (:121-124).
 * Where the reference re-inverts B from scratch every iteration (Invert :402-456, 4m^3 flop) the
 * engine applies the mathematically equal rank-1 (product-form) update to the device-resident
 * (m+1)x(m+1) matrix [[B^-1, x_B], [c_B B^-1, z]] with the same update kernel as the tableau path:
 * same pivot sequence away from ties, objective within 1e-9 relative (DESIGN.md "Revised path").
 *
 * A: m x n row-major structural columns (the slack identity is implicit); c[n]: costs of the
 * standardised MINIMISATION (c = -C for a Max model, :153-154); b[m] >= -1e-9 (:19-21 is the
 * caller's precondition check). */
typedef struct lpx_revised lpx_revised;
int  lpx_revised_create(int m, int n, const double* A, const double* c, const double* b, lpx_revised** out);
void lpx_revised_destroy(lpx_revised* r);
int  lpx_revised_run(lpx_revised* r, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st);
/* Bidx[m]; Nidx[n] in the reference's list order; xB[m]; *z = c_B . x_B of the minimised model. */
int  lpx_revised_result(lpx_revised* r, int32_t* Bidx, int32_t* Nidx, double* xB, double* z);
int  lpx_revised_binv(lpx_revised* r, double* Binv /* [m*m] row-major */);
/* What the reference prints per iteration (BuildIterationBlock, Models/RevisedPrimalSimplex.cs:191-246), as
 * left by the LAST completed iteration: rc[n+m] = reduced cost of every column as priced at its start
 * (+inf for columns that were basic then; rN of :71 is rc gathered in that iteration's Nidx order) and
 * d[m] = B^-1 a_entering (:96).  theta* (:105) equals xB[leaveRow] after the update.  Either may be NULL. */
int  lpx_revised_iteration_view(lpx_revised* r, double* rc, double* d);
/* K7': recompute [[B^-1, x_B], [c_B B^-1, z]] from the current basis with the device Gauss-Jordan below
 * (what the reference does every iteration, :128-133).  lpx_revised_set_refactor(r, k) makes
 * lpx_revised_run do it after every k iterations (0 = only when the drift policy below asks for it, the default). */
int  lpx_revised_refactor(lpx_revised* r);
int  lpx_revised_set_refactor(lpx_revised* r, int every);
/* How lpx_revised_refactor rebuilds B^-1.  0 (default) = the reference's Invert on the device, bit for bit (:402-456).
 * 1 = fast: Newton-Schulz refinement X <- X + X (I - B X) of the maintained inverse, two dense m x m x m contractions on the
 * FP64 matrix cores (v_mfma_f64_16x16x4_f64); it rounds differently from Invert (bar: same pivots, z within 1e-9,
 * |B^-1 B - I| <= 1e-9) and falls back to mode 0 by itself when the maintained inverse is too far off to contract. */
int  lpx_revised_set_refactor_mode(lpx_revised* r, int mode);
/* Drift control of the product-form inverse (the reference never drifts: it re-inverts every iteration).  Every
 * `check_every` iterations lpx_revised_run evaluates two residuals on the device -- rho = max_i |(B x_B)_i - b_i| / (1 + max_i |b_i|)
 * and, for a fixed probe vector v of order one, rho2 = max_i |(B^-1 (B v))_i - v_i| / (1 + max|v|), which sees an error of B^-1 in
 * directions b does not excite (three m x m sweeps in all) -- and refactorises when max(rho, rho2) > tol.  Default: check_every = 256, tol = 1e-9; check_every = 0 switches it off.
 * lpx_revised_set_refactor(r, k > 0) replaces it by an unconditional refactorisation every k iterations. */
int  lpx_revised_set_drift_policy(lpx_revised* r, int check_every, double tol);
int  lpx_revised_residual(lpx_revised* r, double* rel /* max(rho, rho2) */, double* abs_ /* max_i |(B x_B)_i - b_i| */);
/* Counters since creation; gemm_ms / gemm_calls: HIP-event time and number of the matrix-core contractions (2 m^3 flop each)
 * of the LAST fast refactorisation.  Any pointer may be NULL. */
int  lpx_revised_refactor_stats(lpx_revised* r, int* refactors, int* fast_steps, int* fast_fallbacks, double* last_residual,
                                double* gemm_ms, int* gemm_calls);
/* Measurement: mean HIP-event duration [us] of each kernel of the engine's four-launch iteration -- us[0] rv_price (r_N = c_N - pi N,
 * MultiplyRow + Subtract :71-72), us[1] rv_pick (entering candidate :76-83, column copy :95), us[2] rv_upd_ftran (d = B^-1 a_q, Multiply
 * :96, fused with the previous pivot's update of [[B^-1, x_B]]), us[3] rv_select2 (ratio test :99-112, bookkeeping :121-124) -- over up to
 * `iters` real iterations from the handle's current basis; *measured = iterations that completed a pivot. */
int  lpx_revised_profile(lpx_revised* r, int iters, double* us /* [4] */, int* measured);
/* Invert (Models/RevisedPrimalSimplex.cs:402-456), bit for bit: Gauss-Jordan with partial pivoting on
 * [M | I]; M and inv are n x n row-major host buffers.  Returns 0 or LPX_E_SINGULAR (:426). */
int  lpx_invert(const double* M, int n, double* inv);
int  lpx_revised_trace(lpx_revised* r, int32_t* trace /* [2*cap]: (leaveRow, entering) */, int cap, int* n);
/* one-shot on host buffers */
int  lpx_revised_solve(const double* A, int m, int n, const double* c, const double* b,
                       int32_t* Bidx, int32_t* Nidx, double* xB, double* z,
                       double eps, int max_iter, lpx_pivot_cb cb, void* user, lpx_stats* st);

/* ---- 0/1 knapsack branch and bound: batched bounds ------------------------------------------- */
/* Items are kept in HBM in the reference's ratio order (Models/BranchAndBoundKnapsack.cs:75-79).
 * lpx_knapsack_relax_batch evaluates ComputeRelaxation (:431-491) for `count` nodes in one launch, one
 * workgroup per node.  Node k is the list of its fixed decisions fix_idx[off[k]..off[k+1]) (ORIGINAL
 * item indices, ascending) with values fix_val (0/1) -- the reference's int[n] Assigned (:25) without
 * the undecided entries.  Outputs per node: profit (= bound), weight, the fractional item's position in
 * ratio order (-1 if none) and its fraction. */
typedef struct lpx_knapsack lpx_knapsack;
int  lpx_knapsack_create(const double* profit, const double* weight, int n, double cap, lpx_knapsack** out);
void lpx_knapsack_destroy(lpx_knapsack* k);
int  lpx_knapsack_order(lpx_knapsack* k, int32_t* order /* [n]: ratio rank -> original index */);
int  lpx_knapsack_relax_batch(lpx_knapsack* k, int count, const int32_t* off, const int32_t* fix_idx,
                              const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                              double* frac_val);
/* Same, plus each node's two children in the same launch: outputs have 3*count entries, slot 3k = node k itself,
 * 3k+1 / 3k+2 = node k with its fractional item (the one the search branches on, :180) additionally fixed to 0 / 1 --
 * what the best-first loop (:207-209, :267-269) asks for when it pops node k.  frac_idx = -2 in the child slots when
 * node k has no fractional item.  Needs non-negative weights (lpx_knapsack_has_prefix). */
int  lpx_knapsack_relax_batch2(lpx_knapsack* k, int count, const int32_t* off, const int32_t* fix_idx,
                               const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                               double* frac_val);
int  lpx_knapsack_has_prefix(lpx_knapsack* k);
/* Device-resident node store: the best-first loop makes every node as "its parent plus one decision" (:207-209, :267-269), so
 * fixed lists never travel.  Job j derives the node `parent[j]` (-1 = the root, nothing fixed; else an id returned earlier)
 * + `item[j]` fixed to `val[j]`, keeps its list in HBM under the new id child[j], and returns what lpx_knapsack_relax_batch2
 * returns for it: slot 3j the node itself, 3j+1 / 3j+2 the node with its fractional item fixed to 0 / 1 -- whose lists are
 * stored too, under the ids child[j] + 1 / child[j] + 2 (meaningful when frac_idx[3j] >= 0).  48 bytes go to the device per
 * job instead of the whole list.  Needs non-negative weights (lpx_knapsack_has_prefix). */
int  lpx_knapsack_expand_batch(lpx_knapsack* k, int count, const int64_t* parent, const int32_t* item, const int8_t* val,
                               int64_t* child, double* profit, double* weight, int32_t* frac_idx, double* frac_val);
/* The same in two halves (the children of Models/BranchAndBoundKnapsack.cs:207-209,:267-269 evaluated AHEAD of the loop that
 * asks for them), so that the host can work while the device does: _begin enqueues the batch (ids in child[] are valid
 * at once), _finish waits for it and hands out the 3 * count results.  One batch in flight per handle. */
int  lpx_knapsack_expand_begin(lpx_knapsack* k, int count, const int64_t* parent, const int32_t* item, const int8_t* val, int64_t* child);
int  lpx_knapsack_expand_finish(lpx_knapsack* k, double* profit, double* weight, int32_t* frac_idx, double* frac_val);
/* The stored list of a node (ORIGINAL item indices ascending, values 0/1); *depth = its length. */
int  lpx_knapsack_node_list(lpx_knapsack* k, int64_t node, int32_t* idx, int8_t* val, int cap, int* depth);

/* ---- model level: the reference's plugin boundary through a C ABI ------------------------------ */
/* `ILPAlgorithm.Solve(LPProblem, Action<string,bool[,]>) -> SimplexResult` (Models/IPLAlgorithm.cs:5-8)
 * as dispatched by `LPSolver.Solve(problem, algorithmName, cb)` (Models/LPSolver.cs:16-59).  The host
 * side (model preparation, tableau construction, report text, B&B tree) is the C++ mirror in
 * csrc/host/; the loops run on the GPU.  Used by tools/lpx_cli and the Python binding. */
enum { LPX_MAX = 0, LPX_MIN = 1 };                 /* Sense, Models/PrimalSimplex.cs:8 */
enum { LPX_LE = 0, LPX_GE = 1, LPX_EQ = 2 };       /* Rel,   Models/PrimalSimplex.cs:9 */

typedef struct lpx_problem {                       /* LPProblem, Models/PrimalSimplex.cs:20-36 */
    int sense, n, m;
    const double* c;      /* [n]   C */
    const double* A;      /* [m*n] Constraints[i].A, row-major */
    const int32_t* rel;   /* [m]   Constraints[i].Relation */
    const double* b;      /* [m]   Constraints[i].B */
} lpx_problem;

typedef struct lpx_solve_opts {
    int max_iter;          /* 0 = 10000 */
    int batch;             /* 0 = default */
    int render_iterations; /* 1 = text callback receives the whole formatted tableau per pivot */
    int dual_flags;        /* 0 = faithful DualSimplex (defects D1/D2), 7 = repaired */
    int bnb_mode;          /* 0 = faithful, 1 = repaired */
    int bnb_search;        /* 0 = reference DFS, 1 = level-synchronous sharded frontier (every node re-solved from the
                              slack basis, as the reference), 2 = the same with warm-started children */
    int concurrent_nodes;  /* node LPs in flight per GPU (level search) */
    int rank, world;       /* shard of this process (level search / knapsack rounds) */
    int64_t max_nodes;     /* 0 = unlimited; sharded searches: node budget of the WHOLE job, split over the ranks */
    /* incumbent exchange, MAX over ranks in place (RCCL all-reduce in production); NULL = 1 process */
    void (*allreduce_max)(void* user, double* vals, int count);
    void* allreduce_user;
    /* Action<string,bool[,]>: text + optional R x C highlight mask (NULL = none) */
    void (*text_cb)(void* user, const char* text, const uint8_t* highlight, int R, int C);
    void* text_user;
    int bnb_dive;          /* sharded searches: 0 = whole frontier per round (breadth first), 1 = only the deepest
                              `concurrent_nodes` nodes of the pool per round (depth-first-K: reaches incumbents early) */
} lpx_solve_opts;

typedef struct lpx_result {                        /* SimplexResult, Models/PrimalSimplex.cs:38-49 */
    int status;            /* LPX_OPTIMAL / LPX_UNBOUNDED / LPX_INFEASIBLE */
    int has_solution;      /* 0 == Solution/Tableau/Basis/VarNames are null in the reference (defect D2, revised) */
    double optimal_value;  /* OptimalValue */
    int n; double* x;      /* Solution [n] */
    int R, C; double* T;   /* Tableau [R*C] */
    int32_t* basis;        /* Basis [R-1] */
    int n_pivots; int32_t* trace;   /* (row, col) per pivot */
    char* report; char* summary;    /* Report, Summary */
    int64_t lp_solves, nodes;       /* branch and bound */
    int n_log; int32_t* node_log;   /* [3*n_log]: depth, outcome, branching variable */
    double* node_z;                 /* [n_log] */
    double aux[4];                  /* revised: {z_original, z_internal}; knapsack: {relaxations, popped, expanded, max_heap};
                                       sharded B&B: {levels, all-reduces, rebalancing rounds, node descriptors moved} */
    lpx_stats stats;
    int n_cuts; double* cuts;       /* cutting plane: [n_cuts*(nvars+1)] = (A[0..nvars), B) per cut, in the order added */
} lpx_result;

void lpx_default_solve_opts(lpx_solve_opts* o);
/* Returns 0 and fills *out, or the negative LPX_E_* code of the exception the reference would throw
 * (message via lpx_last_error). LPX_ITER_LIMIT (3) is returned for its iteration-limit exceptions. */
int  lpx_solve(const lpx_problem* p, const char* algorithm, const lpx_solve_opts* o, lpx_result* out);
void lpx_result_free(lpx_result* r);

typedef struct lpx_parsed { int sense, n, m; double* c; double* A; int32_t* rel; double* b; int ragged; } lpx_parsed;
/* LPParser.ParseFromText, Models/LPParser.cs:9-79 */
int  lpx_parse_text(const char* text, lpx_parsed* out);
void lpx_parsed_free(lpx_parsed* p);
/* ToString("0.###") as the reference renders tableau cells (Models/PrimalSimplex.cs:280) */
int  lpx_format_number(double v, char* buf, int len);

/* ---- consumers of SimplexResult.Tableau / Basis ------------------------------------
 * CuttingPlane / CuttingPlaneRevised (Models/CuttingPlane.cs:13-139, Models/CuttingPlaneRevised.cs:14-78) are
 * reached through lpx_solve with the menu names Form1.cs:249-261 uses: "Cutting Plane", "Revised Cutting Plane".
 *
 * SensitivityAnalysis (Models/SensitivityAnalysis.cs:11-297) over a problem and the final (T, basis) of a solve of
 * it; VarNames are the reference's x1..xn, c1..cm.  Every call repeats the constructor's checks (:24-43) and
 * returns the negative code with the reference's message in lpx_last_error.  Text is copied into buf (NUL
 * terminated, truncated to len); the return value is the full length.  The reference indexes the tableau as if
 * its objective row came first (:122,:237,:263,:279) -- kept. */
int lpx_sensitivity_range_report(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                 const char* target, char* buf, int len);                  /* GetRangeReport :47-76 */
int lpx_sensitivity_range(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                          const char* target, double* min, double* max);                   /* its numbers, :229-298 */
/* ApplyChange :78-107.  The model belongs to the caller: *field = 0 -> Constraints[*index].B = value,
 * 1 -> C[*index] = value is what the reference would have assigned. */
int lpx_sensitivity_apply_change(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                 const char* target, double value, int* field, int* index, char* buf, int len);
int lpx_sensitivity_shadow_prices(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                  char* buf, int len);                                     /* GetShadowPricesReport :109-128 */
/* SolveUsingDuality :130-219: builds the dual model and runs "Dual Simplex" on it (opts->dual_flags as lpx_solve). */
int lpx_sensitivity_solve_duality(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                  const lpx_solve_opts* o, lpx_result* out);

#ifdef __cplusplus
}
#endif
#endif /* LPX_H */
