/*
 * lpx_oracle.h -- CPU restatement ("oracle") of the reference's simplex / branch-and-bound
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 *   * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *     The product (liblpx.so) never links, loads or calls anything in oracle/.
 *   * Every function follows the reference C# loop order literally (strict '<', the
 *     'best - tol' hysteresis, first-index ties, separate multiply and subtract, true IEEE
 *     division) and cites the reference file:line it restates.  Paths are relative to
 *     /root/reference/Linear_Programming_Solver/.
 *   * Build: gcc -O2 -ffp-contract=off (no -march=native, no -ffast-math) so that
 *     `t -= f*p` rounds twice exactly as RyuJIT's scalar SSE2 code does.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, fixtures, golden vectors
 * or sample files and cannot be compiled here (C# net8.0-windows
 * WinForms, no .NET toolchain in the image).  The oracle is pinned only by hand-derived
 * known-answer tests authored from the cited semantics (tests/golden/kat_*.json) and, for
 * LP optima, by an independent SciPy cross-check.
 */
#ifndef LPX_ORACLE_H
#define LPX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status / error codes (shared numbering with include/lpx.h) ------------------------ */
enum {
    ORC_OPTIMAL      = 0,   /* "OPTIMAL"    Models/PrimalSimplex.cs:126, Models/DualSimplex.cs:73 */
    ORC_UNBOUNDED    = 1,   /* "UNBOUNDED"  Models/PrimalSimplex.cs:104-105                      */
    ORC_INFEASIBLE   = 2,   /* "INFEASIBLE" Models/DualSimplex.cs:94-95                          */
    ORC_ITER_LIMIT   = 3,   /* exception "Iteration limit exceeded." PrimalSimplex.cs:96, DualSimplex.cs:39,
                               RevisedPrimalSimplex.cs:144 */
    /* hard failures the reference raises as System.Exception */
    ORC_E_GE_PRESENT = -10, /* Models/PrimalSimplex.cs:70  */
    ORC_E_NEG_RHS    = -11, /* Models/PrimalSimplex.cs:75  */
    ORC_E_REVISED_PRECOND = -12, /* Models/RevisedPrimalSimplex.cs:21 */
    ORC_E_SINGULAR   = -13, /* Models/RevisedPrimalSimplex.cs:426 */
    ORC_E_KNAP_SHAPE = -14, /* Models/BranchAndBoundKnapsack.cs:66-69 */
    ORC_E_INVAL      = -1
};

enum { ORC_MAX = 0, ORC_MIN = 1 };          /* Sense, Models/PrimalSimplex.cs:8 */
enum { ORC_LE = 0, ORC_GE = 1, ORC_EQ = 2 };/* Rel,   Models/PrimalSimplex.cs:9 */

/* LPProblem (Models/PrimalSimplex.cs:20-36) flattened: A is row-major m x n. */
typedef struct {
    int sense;
    int n;                 /* NumVars */
    int m;                 /* Constraints.Count */
    const double* c;       /* [n] */
    const double* A;       /* [m*n] */
    const int32_t* rel;    /* [m] */
    const double* b;       /* [m] */
} orc_problem;

/* SimplexResult (Models/PrimalSimplex.cs:38-49) without the text fields.
 * has_solution==0 reproduces defect D2 (Models/DualSimplex.cs:310: Solution/Tableau/Basis/VarNames null). */
typedef struct {
    int status;
    int has_solution;
    double z;              /* OptimalValue */
    int n;                 /* number of x entries */
    double* x;             /* Solution [n] */
    int R, C;              /* tableau shape */
    double* T;             /* Tableau [R*C] row-major (final) */
    int32_t* basis;        /* [R-1] */
    int n_pivots;
    int32_t* trace;        /* [2*n_pivots] = (leaving_row, entering_col) per pivot, in order */
    int n_fdf_pivots;      /* dual only: pivots spent in ForceDualFeasibility (first in trace) */
} orc_result;

void orc_result_free(orc_result* r);

/* ---- tableau-level primitives (Models/PrimalSimplex.cs:205-257) -------------------------- */
int  orc_choose_entering(const double* T, int R, int C, double eps);                 /* :205-220 */
int  orc_choose_leaving(const double* T, int R, int C, int q, double eps, double tol); /* :222-243 (tol=eps) ;
                                                       Models/DualSimplex.cs:212-222 (tol=1e-12) */
void orc_pivot(double* T, int R, int C, int r, int q);                               /* :245-257 */

/* Primal loop on a prepared tableau (Models/PrimalSimplex.cs:92-124). trace may be NULL.
 * Returns ORC_OPTIMAL / ORC_UNBOUNDED / ORC_ITER_LIMIT. */
int orc_primal_tableau(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                       int32_t* trace, int* n_pivots);
/* Same loop with the row updates of Pivot spread over `threads` OpenMP threads (0 = all cores):
 * the multi-core courtesy CPU baseline of bench.py; bit-identical results (primal_mt.c). */
int orc_primal_tableau_mt(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                          int32_t* trace, int* n_pivots, int threads);

/* Dual loop on a prepared tableau (Models/DualSimplex.cs:24 + :36-113).
 * fdf_guard = 100 in the reference (:202).  cleanup!=0 adds the repaired-mode primal clean-up
 * phase (see oracle/dual.c header).  Returns ORC_OPTIMAL / ORC_INFEASIBLE / ORC_ITER_LIMIT. */
int orc_dual_tableau(double* T, int R, int C, int32_t* basis, double eps, double ratio_tol,
                     int fdf_guard, int max_iter, int cleanup,
                     int32_t* trace, int* n_pivots, int* n_fdf);

/* Forced pivots for the K4 headline microbenchmark / bitwise kernel parity:
 * for k in [0,count): r = rows[k]; q = first column >= cols[k] (wrapping over [0,C)) with
 * |T[r,q]| >= thresh; pivot(r,q).  chosen[k] receives q (or -1 if none, pivot skipped). */
void orc_forced_pivots(double* T, int R, int C, const int32_t* rows, const int32_t* cols,
                       int count, double thresh, int32_t* chosen);

/* ---- model-level solvers ------------------------------------------------------------------ */
/* PrimalSimplex.Solve, Models/PrimalSimplex.cs:57-127 */
int orc_primal_solve(const orc_problem* p, int max_iter, orc_result* out);

/* DualSimplex.Solve, Models/DualSimplex.cs:15-114.
 * flags: bit0 = repair D1 (no second sign flip, :148-153); bit1 = repair D2 (return Solution/
 * Tableau/Basis); bit2 = lift the ForceDualFeasibility guard (100 -> max_iter) and run the
 * primal clean-up phase.  flags==0 is the faithful reference. */
#define ORC_DUAL_FIX_D1   1
#define ORC_DUAL_FIX_D2   2
#define ORC_DUAL_SOUND    4
#define ORC_DUAL_REPAIRED 7
int orc_dual_solve(const orc_problem* p, int flags, int max_iter, orc_result* out);

/* RevisedPrimalSimplex.Solve, Models/RevisedPrimalSimplex.cs:17-145 (full re-invert per
 * iteration, as the reference).  The reference returns text only; the oracle exposes the
 * numbers the text is rendered from (x, z from ORIGINAL c, :286-292). */
typedef struct {
    int status;
    int n, m;
    double z_original;     /* sum original.C[j]*x[j], :287-289 */
    double z_internal;     /* cB . xB of the minimised standardised model, :133 */
    double* x;             /* [n] */
    int32_t* Bidx;         /* [m] final basis */
    int32_t* Nidx;         /* [n] final nonbasic list in reference order */
    double* xB;            /* [m] */
    int n_iters;
    int32_t* trace;        /* [2*n_iters] = (leaveRow, entering column) */
} orc_revised_result;
void orc_revised_result_free(orc_revised_result* r);
int orc_revised_solve(const orc_problem* p, int max_iter, orc_revised_result* out);
/* Invert, Models/RevisedPrimalSimplex.cs:402-456; M,inv are n x n row-major. returns 0 or ORC_E_SINGULAR */
int orc_invert(const double* M, int n, double* inv);

/* BranchAndBound.Solve, Models/Branch&Bound.cs:30-123.
 * mode 0 = faithful (DualSimplex flags 0 -> every >= child is "Invalid", D1/D2),
 * mode 1 = repaired (DualSimplex flags ORC_DUAL_REPAIRED). */
typedef struct {
    int status;            /* 0 = finished with incumbent, 1 = finished without incumbent,
                              2 = root infeasible/error, 3 = root invalid result (D2) */
    double best_z;
    int n;
    double* best_x;        /* [n] or all zeros when no incumbent */
    int has_incumbent;
    int64_t lp_solves;     /* calls reaching _solver.Solve (:57, :148) */
    int64_t nodes_visited; /* SolveNode entries */
    int64_t total_pivots;
    int max_depth_seen;
    /* node log in visit order (SolveNode entries): outcome code per node */
    int n_log;
    int32_t* log_depth;
    int32_t* log_outcome;  /* see ORC_BNB_* */
    int32_t* log_branch_var; /* fracIndex or -1 */
    double* log_z;
} orc_bnb_result;
enum {
    ORC_BNB_ERROR = 0,      /* exception in solve (:150-154) */
    ORC_BNB_INVALID = 1,    /* missing Solution/Tableau/Basis (:157-161) */
    ORC_BNB_INFEASIBLE_X = 2,/* IsFeasible false (:175-179) */
    ORC_BNB_PRUNED = 3,     /* z <= Best + EPS (:182-186) */
    ORC_BNB_INCUMBENT = 4,  /* integral (:189-195) */
    ORC_BNB_NO_FRAC = 5,    /* :215-219 */
    ORC_BNB_BRANCHED = 6,
    ORC_BNB_DEPTH = 7,      /* :132-136 */
    ORC_BNB_LP_INFEASIBLE = 8 /* repaired mode only: relaxation ended INFEASIBLE */
};
void orc_bnb_result_free(orc_bnb_result* r);
int orc_bnb_solve(const orc_problem* p, int mode, int max_iter, int64_t max_nodes, orc_bnb_result* out);
/* ---- consumers of SimplexResult.Tableau / Basis, consumers.c --------------------- */
enum {
    ORC_CUT_INTEGER = 0,     /* "Status: OPTIMAL INTEGER", CuttingPlane.cs:91-104 / CuttingPlaneRevised.cs:49-57 */
    ORC_CUT_INCOMPLETE = 1,  /* iteration limit 50, CuttingPlane.cs:132-137 / CuttingPlaneRevised.cs:70-77 */
    ORC_CUT_ERROR = 2,       /* solver exception, CuttingPlane.cs:42-50 (CuttingPlaneRevised lets it propagate) */
    ORC_CUT_NONBASIC = 3,    /* "Variable x.. is not basic", CuttingPlane.cs:116-124 */
    ORC_CUT_NOT_OPTIMAL = 4  /* CuttingPlaneRevised.cs:27-35 */
};
typedef struct {
    int status, error;
    int n, n_cuts;
    double* cut_A;           /* [n_cuts*n] cuts in the order they were added */
    double* cut_b;           /* [n_cuts] */
    double* x; double z;     /* last LP solution / objective the loop looked at */
    int64_t lp_solves, total_pivots;
} orc_cut_result;
void orc_cut_result_free(orc_cut_result* r);
int orc_cutting_plane(const orc_problem* p, int max_iter, orc_cut_result* out);          /* Models/CuttingPlane.cs:13-139 */
int orc_cutting_plane_revised(const orc_problem* p, int max_iter, orc_cut_result* out);  /* Models/CuttingPlaneRevised.cs:14-78 */
/* SensitivityAnalysis numbers, Models/SensitivityAnalysis.cs:229-298 and :109-128 (see consumers.c) */
int orc_sens_range(const orc_problem* p, const double* T, int R, int C, const int32_t* basis,
                   int kind, int index, double* pmin, double* pmax, int* which);
void orc_sens_shadow_prices(const orc_problem* p, const double* T, int R, int C, double* shadow);

/* BranchAndBoundRevised.Solve, Models/BranchAndBoundRevised.cs:27-98: same result record;
 * node x*, z* quantised to 3 decimals by the reference's Summary-text round trip. */
int orc_bnbr_solve(const orc_problem* p, int mode, int max_iter, int64_t max_nodes, orc_bnb_result* out);

/* BranchAndBoundKnapsack.Solve, Models/BranchAndBoundKnapsack.cs:58-407 */
typedef struct {
    int status;            /* 0 = BEST CANDIDATE FOUND, 1 = none */
    double best_z;
    int n;
    int32_t* best_x;       /* [n] 0/1 */
    int64_t nodes_popped;  /* nodeCounter, :121 */
    int64_t nodes_expanded;/* pops that passed the bound test (:124) */
    int64_t relaxations;   /* ComputeRelaxation calls */
    int64_t max_heap;
} orc_knap_result;
void orc_knap_result_free(orc_knap_result* r);
int orc_knapsack_solve(const orc_problem* p, int64_t max_nodes, orc_knap_result* out);
/* ComputeRelaxation, :431-491, exposed for kernel parity. order = item indices in ratio order
 * (from orc_knapsack_order). Outputs bound(profit), weight, fractional sorted idx. */
void orc_knapsack_order(const double* profit, const double* weight, int n, int32_t* order);
void orc_knapsack_relax(const double* profit, const double* weight, int n, double cap,
                        const int32_t* order, const int32_t* assigned,
                        double* relaxed /*[n] or NULL*/, double* out_profit, double* out_weight,
                        int32_t* out_frac_sorted_idx);

/* LPParser.ParseFromText, Models/LPParser.cs:9-79.  Returns 0 or a negative error; on success
 * fills malloc'ed arrays that the caller releases with orc_parsed_free. */
typedef struct { int sense, n, m; double* c; double* A; int32_t* rel; double* b; int ragged; } orc_parsed;
int  orc_parse_text(const char* text, orc_parsed* out, char* err, int errlen);
void orc_parsed_free(orc_parsed* p);

#ifdef __cplusplus
}
#endif
#endif
