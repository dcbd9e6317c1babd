/*
 * oracle/bnbr.c -- CPU restatement of Models/BranchAndBoundRevised.cs (TEST INFRASTRUCTURE, see
 * lpx_oracle.h).  Same DFS as BranchAndBound, but every node is solved by RevisedPrimalSimplex (all
 * `<=`) or DualSimplex (any `>=`/`=`) and x*, z* are READ BACK FROM THE SUMMARY TEXT (:276-389), i.e.
 * quantised to three decimals by `Math.Round(v, 3)` (Models/RevisedPrimalSimplex.cs:284,292,
 * Models/DualSimplex.cs:307-308).  Because the text exists even when DualSimplex returns no arrays
 * (defect D2), `>=` children are NOT dropped here; defect D1 still rewrites their constraint.
 * The oracle skips the text and applies the same rounding to the numbers (print + parse of a value
 * already rounded to 3 decimals is the identity).
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define EPS 1e-6        /* :21 */
#define MAXDEPTH 200    /* :22 */

typedef struct { int var; int rel; double bound; } cut_t;
typedef struct {
    const orc_problem* root; int mode; int max_iter; int64_t max_nodes;
    double best; int has_best; double* best_x; orc_bnb_result* out; int cap; int stop;
} ctx_t;

/* Math.Round(value, 3): banker's rounding of value*1000 (BCL, |value| < 1e16) */
static double round3(double v) { return fabs(v) < 1e16 ? nearbyint(v * 1000.0) / 1000.0 : v; }

static void log_node(ctx_t* c, int depth, int outcome, int var, double z)
{
    orc_bnb_result* o = c->out;
    if (o->n_log == c->cap) {
        c->cap = c->cap ? c->cap * 2 : 256;
        o->log_depth = (int32_t*)realloc(o->log_depth, sizeof(int32_t) * c->cap);
        o->log_outcome = (int32_t*)realloc(o->log_outcome, sizeof(int32_t) * c->cap);
        o->log_branch_var = (int32_t*)realloc(o->log_branch_var, sizeof(int32_t) * c->cap);
        o->log_z = (double*)realloc(o->log_z, sizeof(double) * c->cap);
    }
    o->log_depth[o->n_log] = depth; o->log_outcome[o->n_log] = outcome;
    o->log_branch_var[o->n_log] = var; o->log_z[o->n_log] = z;
    o->n_log++;
}

static void build_node(const ctx_t* c, const cut_t* cuts, int nc, orc_problem* np, double** pA, int32_t** prel, double** pb)
{
    const orc_problem* r = c->root;
    int n = r->n, m = r->m + nc;
    double* A = (double*)calloc((size_t)m * n, sizeof(double));
    int32_t* rel = (int32_t*)malloc(sizeof(int32_t) * m);
    double* b = (double*)malloc(sizeof(double) * m);
    memcpy(A, r->A, sizeof(double) * (size_t)r->m * n);
    memcpy(rel, r->rel, sizeof(int32_t) * r->m);
    memcpy(b, r->b, sizeof(double) * r->m);
    for (int k = 0; k < nc; k++) { A[(size_t)(r->m + k) * n + cuts[k].var] = 1.0; rel[r->m + k] = cuts[k].rel; b[r->m + k] = cuts[k].bound; }
    np->sense = r->sense; np->n = n; np->m = m; np->c = r->c; np->A = A; np->rel = rel; np->b = b;
    *pA = A; *prel = rel; *pb = b;
}

/* LP relaxation + the text round trip (:120-147).  returns 0 ok, 1 error (exception), 2 invalid z */
static int solve_and_parse(ctx_t* c, const orc_problem* p, double* x, double* z)
{
    int n = p->n, ge = 0;
    for (int i = 0; i < p->m; i++) if (p->rel[i] == ORC_GE || p->rel[i] == ORC_EQ) { ge = 1; break; }   /* :238-243 */
    c->out->lp_solves++;
    if (!ge) {
        orc_revised_result r;
        int st = orc_revised_solve(p, c->max_iter, &r);
        if (st < 0 || st == ORC_ITER_LIMIT) { if (st != ORC_E_REVISED_PRECOND) orc_revised_result_free(&r); return 1; }
        c->out->total_pivots += r.n_iters;
        for (int j = 0; j < n; j++) x[j] = round3(r.x[j]);
        *z = round3(r.z_original);
        orc_revised_result_free(&r);
    } else {
        orc_result r;
        int flags = c->mode ? ORC_DUAL_REPAIRED : ORC_DUAL_FIX_D2;   /* numbers needed either way; D1 kept in faithful mode */
        int st = orc_dual_solve(p, flags, c->max_iter, &r);
        if (st < 0 || st == ORC_ITER_LIMIT) { orc_result_free(&r); return 1; }
        c->out->total_pivots += r.n_pivots;
        for (int j = 0; j < n; j++) x[j] = round3(r.x[j]);
        *z = round3(r.z);
        orc_result_free(&r);
    }
    /* all |x| < EPS -> ParseSolutionVectorFromTableau finds no "xN <rhs>" line in a Summary -> zeros (:327-369) */
    int allz = 1;
    for (int j = 0; j < n; j++) if (!(fabs(x[j]) < EPS)) { allz = 0; break; }
    if (allz) for (int j = 0; j < n; j++) x[j] = 0.0;
    if (isnan(*z) || isinf(*z)) return 2;                              /* :143-147 */
    return 0;
}

static int is_integral(const double* x, int n)                        /* :245-248: strict '<' */
{
    for (int i = 0; i < n; i++) if (!(fabs(x[i] - rint(x[i])) < EPS)) return 0;
    return 1;
}

static int is_feasible(const double* x, const orc_problem* p)         /* :250-266 */
{
    int n = p->n;
    for (int k = 0; k < p->m; k++) {
        const double* a = p->A + (size_t)k * n;
        double sum = 0;
        for (int i = 0; i < n; i++) sum += a[i] * x[i];
        if (p->rel[k] == ORC_LE && sum > p->b[k] + EPS) return 0;
        if (p->rel[k] == ORC_GE && sum < p->b[k] - EPS) return 0;
        if (p->rel[k] == ORC_EQ && fabs(sum - p->b[k]) > EPS) return 0;
    }
    for (int i = 0; i < n; i++) if (!(x[i] >= -EPS)) return 0;
    return 1;
}

static void solve_sub(ctx_t* c, cut_t* cuts, int nc, int depth)       /* SolveSubproblem, :100-234 */
{
    if (c->stop) return;
    if (c->max_nodes > 0 && c->out->nodes_visited >= c->max_nodes) { c->stop = 1; return; }
    c->out->nodes_visited++;
    if (depth > c->out->max_depth_seen) c->out->max_depth_seen = depth;
    if (depth > MAXDEPTH) { log_node(c, depth, ORC_BNB_DEPTH, -1, 0.0); return; }
    int n = c->root->n;
    orc_problem np; double* A; int32_t* rel; double* b;
    build_node(c, cuts, nc, &np, &A, &rel, &b);
    double* x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    double z = 0;
    int rc = solve_and_parse(c, &np, x, &z);
    int k = -1; double fl = 0, ce = 0;
    if (rc == 1) log_node(c, depth, ORC_BNB_ERROR, -1, 0.0);
    else if (rc == 2) log_node(c, depth, ORC_BNB_INVALID, -1, z);
    else if (!is_feasible(x, &np)) log_node(c, depth, ORC_BNB_INFEASIBLE_X, -1, z);              /* :152-156 */
    else if (z <= (c->has_best ? c->best : -INFINITY) + EPS) log_node(c, depth, ORC_BNB_PRUNED, -1, z);   /* :159-163 */
    else if (is_integral(x, n)) {                                                                 /* :166-172 */
        c->best = z; c->has_best = 1;
        for (int i = 0; i < n; i++) c->best_x[i] = rint(x[i]);
        log_node(c, depth, ORC_BNB_INCUMBENT, -1, z);
    } else {
        double minDist = 1.7976931348623157e308;                                                  /* :175-190 */
        for (int i = 0; i < n; i++) {
            double fp = x[i] - floor(x[i]);
            if (fp > EPS && (1 - fp) > EPS) {
                double dist = fabs(fp - 0.5);
                if (dist < minDist || (dist == minDist && i < k)) { minDist = dist; k = i; }
            }
        }
        if (k == -1) log_node(c, depth, ORC_BNB_NO_FRAC, -1, z);
        else { fl = floor(x[k]); ce = ceil(x[k]); log_node(c, depth, ORC_BNB_BRANCHED, k, z); }   /* :198-200 */
    }
    free(x); free(A); free(rel); free(b);
    if (k < 0) return;
    cuts[nc].var = k; cuts[nc].rel = ORC_GE; cuts[nc].bound = ce;
    solve_sub(c, cuts, nc + 1, depth + 1);                                                        /* ceil first, :232 */
    cuts[nc].var = k; cuts[nc].rel = ORC_LE; cuts[nc].bound = fl;
    solve_sub(c, cuts, nc + 1, depth + 1);                                                        /* :233 */
}

/* BranchAndBoundRevised.Solve, :27-98 */
int orc_bnbr_solve(const orc_problem* p, int mode, int max_iter, int64_t max_nodes, orc_bnb_result* out)
{
    memset(out, 0, sizeof(*out));
    ctx_t c; memset(&c, 0, sizeof(c));
    c.root = p; c.mode = mode; c.max_iter = max_iter; c.max_nodes = max_nodes; c.out = out;
    int n = p->n;
    c.best_x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    out->n = n; out->best_x = c.best_x;
    double* x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    double z = 0;
    int rc = solve_and_parse(&c, p, x, &z);                            /* :40-66 */
    if (rc == 1) { out->status = 2; free(x); return 0; }
    if (is_integral(x, n) && is_feasible(x, p)) {                      /* :69-74 */
        c.best = z; c.has_best = 1;
        for (int i = 0; i < n; i++) c.best_x[i] = rint(x[i]);
    } else {
        cut_t* cuts = (cut_t*)malloc(sizeof(cut_t) * (MAXDEPTH + 4));
        solve_sub(&c, cuts, 0, 0);                                     /* :78 */
        free(cuts);
    }
    free(x);
    out->has_incumbent = c.has_best;
    out->best_z = c.has_best ? c.best : -INFINITY;
    out->status = c.stop ? 4 : (c.has_best ? 0 : 1);
    return 0;
}
