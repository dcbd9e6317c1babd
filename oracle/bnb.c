/*
 * oracle/bnb.c -- CPU restatement of Models/Branch&Bound.cs (class BranchAndBound) (TEST
 * INFRASTRUCTURE, see lpx_oracle.h).  Recursive DFS, ceil child first, every node re-solved
 * from the slack basis through LPSolver -> PrimalSimplex / DualSimplex.
 *
 * mode 0 (faithful): DualSimplex keeps defects D1/D2, so every `>=` child comes back without
 *   Solution/Tableau/Basis and is discarded as "Invalid" (:157-161); the tree is a chain of
 *   floor children.
 * mode 1 (repaired): DualSimplex runs with ORC_DUAL_REPAIRED (oracle/dual.c) and a node whose
 *   relaxation ends INFEASIBLE is pruned explicitly.
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define BNB_EPS 1e-6        /* :24 */
#define BNB_MAXDEPTH 200    /* :25 */

typedef struct {
    int n;
    int m0;                 /* root rows */
    const orc_problem* root;
    int mode;
    int max_iter;
    int64_t max_nodes;
    double best; int has_best; double* best_x;
    orc_bnb_result* out;
    int cap;
    int stop;
} bnb_ctx;

typedef struct { int var; int rel; double bound; } bnb_cut;

void orc_bnb_result_free(orc_bnb_result* r)
{
    if (!r) return;
    free(r->best_x); free(r->log_depth); free(r->log_outcome); free(r->log_branch_var); free(r->log_z);
    memset(r, 0, sizeof(*r));
}

static void log_node(bnb_ctx* c, int depth, int outcome, int var, double z)
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

/* materialise root + cuts as a dense problem (LPProblem.Clone + Constraints.Add, :233-248) */
static void build_node(const bnb_ctx* c, const bnb_cut* cuts, int ncuts, orc_problem* np,
                       double** pA, int32_t** prel, double** pb)
{
    const orc_problem* r = c->root;
    int n = r->n, m = r->m + ncuts;
    double* A = (double*)calloc((size_t)m * n, sizeof(double));
    int32_t* rel = (int32_t*)malloc(sizeof(int32_t) * m);
    double* b = (double*)malloc(sizeof(double) * m);
    memcpy(A, r->A, sizeof(double) * (size_t)r->m * n);
    memcpy(rel, r->rel, sizeof(int32_t) * r->m);
    memcpy(b, r->b, sizeof(double) * r->m);
    for (int k = 0; k < ncuts; k++) {
        A[(size_t)(r->m + k) * n + cuts[k].var] = 1.0;     /* UnitVector, :298-303 */
        rel[r->m + k] = cuts[k].rel;
        b[r->m + k] = cuts[k].bound;
    }
    np->sense = r->sense; np->n = n; np->m = m; np->c = r->c; np->A = A; np->rel = rel; np->b = b;
    *pA = A; *prel = rel; *pb = b;
}

/* ChooseAlgorithm (:262-266) + LPSolver.Solve (Models/LPSolver.cs:16-59) */
static int solve_lp(bnb_ctx* c, const orc_problem* p, orc_result* res)
{
    int hasGEorEQ = 0;
    for (int i = 0; i < p->m; i++) if (p->rel[i] == ORC_GE || p->rel[i] == ORC_EQ) { hasGEorEQ = 1; break; }
    c->out->lp_solves++;
    int st;
    if (hasGEorEQ) st = orc_dual_solve(p, c->mode ? ORC_DUAL_REPAIRED : 0, c->max_iter, res);
    else           st = orc_primal_solve(p, c->max_iter, res);
    c->out->total_pivots += res->n_pivots;
    return st;
}

/* IsIntegral, :268-274 (Math.Round = round-half-to-even = rint in the default mode) */
static int is_integral(const double* x, int n)
{
    for (int i = 0; i < n; i++)
        if (fabs(x[i] - rint(x[i])) > BNB_EPS) return 0;
    return 1;
}

/* IsFeasible, :276-294 */
static int is_feasible(const double* x, const orc_problem* p)
{
    int n = p->n;
    for (int k = 0; k < p->m; k++) {
        const double* a = p->A + (size_t)k * n;
        double sum = 0;
        for (int i = 0; i < n; i++) sum += a[i] * x[i];
        if (p->rel[k] == ORC_LE && sum > p->b[k] + BNB_EPS) return 0;
        if (p->rel[k] == ORC_GE && sum < p->b[k] - BNB_EPS) return 0;
        if (p->rel[k] == ORC_EQ && fabs(sum - p->b[k]) > BNB_EPS) return 0;
    }
    for (int i = 0; i < n; i++) if (x[i] < -BNB_EPS) return 0;
    return 1;
}

static void set_incumbent(bnb_ctx* c, const double* x, double z)
{
    c->best = z; c->has_best = 1;
    for (int i = 0; i < c->n; i++) c->best_x[i] = rint(x[i]);    /* RoundInt, :296 */
}

/* SolveNode, :128-258 */
static void solve_node(bnb_ctx* c, bnb_cut* cuts, int ncuts, int depth)
{
    if (c->stop) return;
    if (c->max_nodes > 0 && c->out->nodes_visited >= c->max_nodes) { c->stop = 1; return; }
    c->out->nodes_visited++;
    if (depth > c->out->max_depth_seen) c->out->max_depth_seen = depth;
    if (depth > BNB_MAXDEPTH) { log_node(c, depth, ORC_BNB_DEPTH, -1, 0.0); return; }   /* :132-136 */

    orc_problem np; double* A; int32_t* rel; double* b;
    build_node(c, cuts, ncuts, &np, &A, &rel, &b);
    orc_result res;
    int st = solve_lp(c, &np, &res);                                      /* :142-154 */
    int n = c->n;
    if (st < 0 || st == ORC_ITER_LIMIT) {                                 /* exception path */
        log_node(c, depth, ORC_BNB_ERROR, -1, 0.0);
        goto cleanup;
    }
    if (!res.has_solution) {                                              /* :157-161 */
        log_node(c, depth, ORC_BNB_INVALID, -1, 0.0);
        goto cleanup;
    }
    if (c->mode == 1 && st == ORC_INFEASIBLE) {
        log_node(c, depth, ORC_BNB_LP_INFEASIBLE, -1, res.z);
        goto cleanup;
    }
    {
        const double* x = res.x;                                          /* :164 */
        double z = res.z;                                                 /* :170 */
        if (!is_feasible(x, &np)) { log_node(c, depth, ORC_BNB_INFEASIBLE_X, -1, z); goto cleanup; }  /* :175-179 */
        double bestObj = c->has_best ? c->best : -INFINITY;
        if (z <= bestObj + BNB_EPS) { log_node(c, depth, ORC_BNB_PRUNED, -1, z); goto cleanup; }      /* :182-186 */
        if (is_integral(x, n)) {                                          /* :189-195 */
            set_incumbent(c, x, z);
            log_node(c, depth, ORC_BNB_INCUMBENT, -1, z);
            goto cleanup;
        }
        int fracIndex = -1;                                               /* :198-213 */
        double minDist = 1.7976931348623157e308;                          /* double.MaxValue */
        for (int i = 0; i < n; i++) {
            double fracPart = x[i] - floor(x[i]);
            if (fracPart > BNB_EPS && (1 - fracPart) > BNB_EPS) {
                double dist = fabs(fracPart - 0.5);
                if (dist < minDist || (dist == minDist && i < fracIndex)) { minDist = dist; fracIndex = i; }
            }
        }
        if (fracIndex == -1) { log_node(c, depth, ORC_BNB_NO_FRAC, -1, z); goto cleanup; }           /* :215-219 */
        double fracVal = x[fracIndex];
        int floorVal = (int)floor(fracVal);                               /* :222 */
        int ceilVal = (int)ceil(fracVal);                                 /* :223 */
        log_node(c, depth, ORC_BNB_BRANCHED, fracIndex, z);
        orc_result_free(&res);
        free(A); free(rel); free(b);
        cuts[ncuts].var = fracIndex; cuts[ncuts].rel = ORC_GE; cuts[ncuts].bound = ceilVal;
        solve_node(c, cuts, ncuts + 1, depth + 1);                        /* ceil first, :256 */
        cuts[ncuts].var = fracIndex; cuts[ncuts].rel = ORC_LE; cuts[ncuts].bound = floorVal;
        solve_node(c, cuts, ncuts + 1, depth + 1);                        /* :257 */
        return;
    }
cleanup:
    orc_result_free(&res);
    free(A); free(rel); free(b);
}

/* BranchAndBound.Solve, :30-123 */
int orc_bnb_solve(const orc_problem* p, int mode, int max_iter, int64_t max_nodes, orc_bnb_result* out)
{
    memset(out, 0, sizeof(*out));
    bnb_ctx c; memset(&c, 0, sizeof(c));
    c.n = p->n; c.m0 = p->m; c.root = p; c.mode = mode; c.max_iter = max_iter; c.max_nodes = max_nodes;
    c.out = out;
    c.best_x = (double*)calloc(p->n > 0 ? p->n : 1, sizeof(double));
    out->n = p->n; out->best_x = c.best_x;

    orc_result root;
    int st = solve_lp(&c, p, &root);                                      /* :50-63 */
    if (st < 0 || st == ORC_ITER_LIMIT) { out->status = 2; orc_result_free(&root); return 0; }
    if (!root.has_solution) { out->status = 3; orc_result_free(&root); return 0; }      /* :66-70 */
    if (is_integral(root.x, p->n) && is_feasible(root.x, p)) {            /* :85-91 */
        set_incumbent(&c, root.x, root.z);
        orc_result_free(&root);
        out->best_z = c.best; out->has_incumbent = 1; out->status = 0;
        return 0;
    }
    orc_result_free(&root);
    bnb_cut* cuts = (bnb_cut*)malloc(sizeof(bnb_cut) * (BNB_MAXDEPTH + 4));
    solve_node(&c, cuts, 0, 0);                                           /* :95 (root solved again) */
    free(cuts);
    out->has_incumbent = c.has_best;
    out->best_z = c.has_best ? c.best : -INFINITY;
    out->status = c.stop ? 4 : (c.has_best ? 0 : 1);
    return 0;
}
