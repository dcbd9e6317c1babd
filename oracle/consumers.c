/*
 * consumers.c -- CPU restatement of the consumers of SimplexResult.Tableau/Basis:
 * CuttingPlane, CuttingPlaneRevised and SensitivityAnalysis.  TEST INFRASTRUCTURE ONLY (lpx_oracle.h).
 *
 * The reference's loops are followed literally, defects included:
 *   * CuttingPlane reads tableau row `i + 1` for the basic variable of row i ("+1 because row 0 is
 *     objective", Models/CuttingPlane.cs:109) although BuildTableau puts the objective row LAST
 *     (Models/PrimalSimplex.cs:197-199), and adds the cut as  sum f_j x_j <= f_0  (:141-162).
 *   * SensitivityAnalysis reads row 0 as the objective row and row index+1 as constraint `index`
 *     (Models/SensitivityAnalysis.cs:122,236,258,262,280) -- same off-by-one.
 *   * CuttingPlaneRevised works on the 3-decimal x* it parses back from the Summary text
 *     (Models/CuttingPlaneRevised.cs:90-110).
 * PARITY UNPINNED BY THE REFERENCE (no fixtures); pinned by the hand-derived KATs in tests/.
 */
#include "lpx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static double round3(double v) { return fabs(v) < 1e16 ? nearbyint(v * 1000.0) / 1000.0 : v; }

static int find_fractional(const double* x, int n)          /* CuttingPlane.cs:76-89, CuttingPlaneRevised.cs:80-88 */
{
    for (int i = 0; i < n; i++) {
        double frac = x[i] - floor(x[i]);
        if (frac > 1e-9 && frac < 1 - 1e-9) return i;
    }
    return -1;
}

void orc_cut_result_free(orc_cut_result* r)
{
    if (!r) return;
    free(r->cut_A); free(r->cut_b); free(r->x);
    memset(r, 0, sizeof(*r));
}

/* One growing model: rows [0, m0) from p, then the cuts. */
typedef struct { int n, m, cap; double* A; int32_t* rel; double* b; } grow_model;

static void grow_init(grow_model* g, const orc_problem* p, int extra)
{
    g->n = p->n; g->m = p->m; g->cap = p->m + extra;
    g->A = (double*)calloc((size_t)g->cap * (p->n > 0 ? p->n : 1), sizeof(double));
    g->rel = (int32_t*)calloc((size_t)g->cap, sizeof(int32_t));
    g->b = (double*)calloc((size_t)g->cap, sizeof(double));
    memcpy(g->A, p->A, sizeof(double) * (size_t)p->m * p->n);
    memcpy(g->rel, p->rel, sizeof(int32_t) * (size_t)p->m);
    memcpy(g->b, p->b, sizeof(double) * (size_t)p->m);
}

/* CuttingPlane.Solve, Models/CuttingPlane.cs:13-139 */
int orc_cutting_plane(const orc_problem* p, int max_iter, orc_cut_result* out)
{
    memset(out, 0, sizeof(*out));
    const int n = p->n, maxIterations = 50;                                 /* :19 */
    grow_model g; grow_init(&g, p, maxIterations);
    out->n = n;
    out->cut_A = (double*)calloc((size_t)maxIterations * (n > 0 ? n : 1), sizeof(double));
    out->cut_b = (double*)calloc((size_t)maxIterations, sizeof(double));
    out->x = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    out->status = ORC_CUT_INCOMPLETE;                                        /* :132-137 */
    for (int iteration = 1; iteration <= maxIterations; iteration++) {       /* :32 */
        orc_problem q = { p->sense, n, g.m, p->c, g.A, g.rel, g.b };
        orc_result r;
        int rc = orc_primal_solve(&q, max_iter, &r);                         /* :40 */
        out->lp_solves++;
        if (rc < 0 || rc == ORC_ITER_LIMIT) {                                /* catch, :42-50 */
            out->status = ORC_CUT_ERROR; out->error = rc;
            orc_result_free(&r);
            break;
        }
        out->total_pivots += r.n_pivots;
        for (int j = 0; j < n; j++) out->x[j] = r.x[j];                      /* :65 */
        out->z = r.z;
        int fracIndex = find_fractional(out->x, n);                          /* :76-89 */
        if (fracIndex == -1) {                                               /* :91-104 */
            out->status = ORC_CUT_INTEGER;
            orc_result_free(&r);
            break;
        }
        int row = -1;                                                        /* :107-115 */
        for (int i = 0; i < r.R - 1; i++)
            if (r.basis[i] == fracIndex) { row = i + 1; break; }
        if (row == -1) {                                                     /* :116-124 */
            out->status = ORC_CUT_NONBASIC;
            orc_result_free(&r);
            break;
        }
        /* GenerateGomoryCut, :141-162 */
        const double* trow = r.T + (size_t)row * r.C;
        double rhs = trow[r.C - 1];
        double f0 = rhs - floor(rhs);
        double* ca = g.A + (size_t)g.m * n;
        for (int j = 0; j < n; j++) {
            double aij = trow[j];
            double fj = aij - floor(aij);
            ca[j] = fj > 1e-9 ? fj : 0.0;
        }
        g.rel[g.m] = ORC_LE; g.b[g.m] = f0;
        memcpy(out->cut_A + (size_t)out->n_cuts * n, ca, sizeof(double) * n);
        out->cut_b[out->n_cuts] = f0;
        out->n_cuts++;
        g.m++;                                                               /* :128 */
        orc_result_free(&r);
    }
    free(g.A); free(g.rel); free(g.b);
    return 0;
}

/* CuttingPlaneRevised.Solve, Models/CuttingPlaneRevised.cs:14-78 */
int orc_cutting_plane_revised(const orc_problem* p, int max_iter, orc_cut_result* out)
{
    memset(out, 0, sizeof(*out));
    const int n = p->n, MaxIterations = 50;                                  /* :12 */
    grow_model g; grow_init(&g, p, MaxIterations);
    out->n = n;
    out->cut_A = (double*)calloc((size_t)MaxIterations * (n > 0 ? n : 1), sizeof(double));
    out->cut_b = (double*)calloc((size_t)MaxIterations, sizeof(double));
    out->x = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    int iter = 1;
    for (;;) {                                                               /* :21 */
        orc_problem q = { p->sense, n, g.m, p->c, g.A, g.rel, g.b };
        orc_revised_result r;
        int rc = orc_revised_solve(&q, max_iter, &r);                        /* :23 (exceptions propagate) */
        out->lp_solves++;
        if (rc < 0 || rc == ORC_ITER_LIMIT) {
            out->status = ORC_CUT_ERROR; out->error = rc;
            if (rc != ORC_E_REVISED_PRECOND) orc_revised_result_free(&r);
            break;
        }
        out->total_pivots += r.n_iters;
        if (rc != ORC_OPTIMAL) {                                             /* :27-35 */
            out->status = ORC_CUT_NOT_OPTIMAL;
            orc_revised_result_free(&r);
            break;
        }
        for (int j = 0; j < n; j++) out->x[j] = round3(r.x[j]);              /* ExtractSolution of the 3-dp text, :37,:90-110 */
        out->z = round3(r.z_original);
        orc_revised_result_free(&r);
        int fracIndex = find_fractional(out->x, n);                          /* :48 */
        if (fracIndex == -1) { out->status = ORC_CUT_INTEGER; break; }       /* :49-57 */
        double floorVal = floor(out->x[fracIndex] + 1e-12);                  /* :59 */
        double* ca = g.A + (size_t)g.m * n;
        for (int j = 0; j < n; j++) ca[j] = j == fracIndex ? 1.0 : 0.0;      /* :60-65 */
        g.rel[g.m] = ORC_LE; g.b[g.m] = floorVal;
        memcpy(out->cut_A + (size_t)out->n_cuts * n, ca, sizeof(double) * n);
        out->cut_b[out->n_cuts] = floorVal;
        out->n_cuts++;
        g.m++;                                                               /* :66 */
        iter++;
        if (iter > MaxIterations) { out->status = ORC_CUT_INCOMPLETE; break; }   /* :70-77 */
    }
    free(g.A); free(g.rel); free(g.b);
    return 0;
}

/* ---- SensitivityAnalysis, Models/SensitivityAnalysis.cs ----------------------------------------- */
static int basis_contains(const int32_t* basis, int m, int col)
{
    for (int i = 0; i < m; i++) if (basis[i] == col) return 1;
    return 0;
}

/* GetRangeReport's numeric part (:47-76).  kind 0: target "Constraint <index+1>" -> GetConstraintRange
 * (:277-298); kind 1: target = variable column `index` -> GetBasicVariableObjectiveRange (:250-275) when the
 * column is basic, else GetNonBasicVariableRange (:229-248).  *which = 0 constraint, 1 basic, 2 non-basic.
 * Returns 0, or ORC_E_INVAL where the reference would throw IndexOutOfRange (problem.C[col] with col >= n). */
int orc_sens_range(const orc_problem* p, const double* T, int R, int C, const int32_t* basis,
                   int kind, int index, double* pmin, double* pmax, int* which)
{
    const int m = R - 1, n = C - 1;
    double mn = -INFINITY, mx = INFINITY;
    if (kind == 0) {
        const int row = index + 1;                                           /* :279 */
        const double currentB = T[(size_t)row * C + n];                      /* :281 */
        for (int j = 0; j < p->n; j++) {                                     /* :286-296 */
            if (basis_contains(basis, m, j)) continue;
            double aij = T[(size_t)row * C + j];
            if (fabs(aij) < 1e-9) continue;
            double delta = -T[(size_t)row * C + n] / aij;
            if (aij > 0) mx = fmin(mx, currentB + delta);
            else mn = fmax(mn, currentB + delta);
        }
        *which = 0;
    } else if (basis_contains(basis, m, index)) {
        int basicVarRow = 0;
        while (basis[basicVarRow] != index) basicVarRow++;                   /* Array.IndexOf, :68 */
        const int col = basis[basicVarRow];
        if (col >= p->n) return ORC_E_INVAL;                                 /* problem.C[col], :256 */
        const double current = p->c[col];
        for (int j = 0; j < n; j++) {                                        /* :260-272 */
            if (basis_contains(basis, m, j)) continue;
            double aij = T[(size_t)(basicVarRow + 1) * C + j];               /* :263 */
            if (fabs(aij) < 1e-9) continue;
            double reducedCost = T[j];                                       /* tableau[0, j], :266 */
            double delta = -reducedCost / aij;
            if (aij > 0) mx = fmin(mx, current + delta);
            else mn = fmax(mn, current + delta);
        }
        *which = 1;
    } else {
        if (index >= p->n) return ORC_E_INVAL;                               /* problem.C[col], :236 */
        const double current = p->c[index];
        const double reducedCost = T[index];                                 /* tableau[0, col], :237 */
        if (reducedCost > 0) mx = current + reducedCost;                     /* :242-245 */
        else if (reducedCost < 0) mn = current + reducedCost;
        *which = 2;
    }
    *pmin = mn; *pmax = mx;
    return 0;
}

/* GetShadowPricesReport's numbers (:109-128): shadow[i] = -tableau[0, nVars + i]. */
void orc_sens_shadow_prices(const orc_problem* p, const double* T, int R, int C, double* shadow)
{
    for (int i = 0; i < p->m; i++) shadow[i] = -T[p->n + i];
}
