/*
 * oracle/primal.c -- CPU restatement of Models/PrimalSimplex.cs (TEST INFRASTRUCTURE, see
 * lpx_oracle.h).  Loop order, comparison forms and rounding follow the C# literally.
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

void orc_result_free(orc_result* r)
{
    if (!r) return;
    free(r->x); free(r->T); free(r->basis); free(r->trace);
    memset(r, 0, sizeof(*r));
}

/* ChooseEntering, Models/PrimalSimplex.cs:205-220: first index of the strict minimum of the
 * LAST row over columns [0, C-1) if below -eps, else -1. */
int orc_choose_entering(const double* T, int R, int C, double eps)
{
    int m = R - 1;
    int cols = C - 1;
    int best = -1;
    double minVal = -eps;
    const double* z = T + (size_t)m * C;
    for (int j = 0; j < cols; j++) {
        if (z[j] < minVal) { minVal = z[j]; best = j; }
    }
    return best;
}

/* ChooseLeaving, Models/PrimalSimplex.cs:222-243.  Sequential hysteresis scan:
 * accept row i iff ratio < bestRatio - tol.  PrimalSimplex uses tol = Eps = 1e-9 (:235);
 * DualSimplex.ForceDualFeasibility uses tol = 1e-12 (Models/DualSimplex.cs:220). */
int orc_choose_leaving(const double* T, int R, int C, int q, double eps, double tol)
{
    int m = R - 1;
    int rhs = C - 1;
    double bestRatio = INFINITY;
    int bestRow = -1;
    for (int i = 0; i < m; i++) {
        double aij = T[(size_t)i * C + q];
        if (aij > eps) {
            double ratio = T[(size_t)i * C + rhs] / aij;
            if (ratio < bestRatio - tol) { bestRatio = ratio; bestRow = i; }
        }
    }
    return bestRow;
}

/* Pivot, Models/PrimalSimplex.cs:245-257 (== Models/DualSimplex.cs:232-246): normalise the
 * pivot row by true division, then for every other row (objective row included) and every
 * column (RHS included) T[i,j] -= factor * T[row,j] with separate multiply and subtract. */
void orc_pivot(double* T, int R, int C, int r, int q)
{
    double* pr = T + (size_t)r * C;
    double piv = pr[q];
    for (int j = 0; j < C; j++) pr[j] /= piv;
    for (int i = 0; i < R; i++) {
        if (i == r) continue;
        double* ti = T + (size_t)i * C;
        double factor = ti[q];
        for (int j = 0; j < C; j++) {
            double prod = factor * pr[j];   /* -ffp-contract=off keeps mul and sub separate */
            ti[j] = ti[j] - prod;
        }
    }
}

/* The while(true) loop of PrimalSimplex.Solve, Models/PrimalSimplex.cs:92-124. */
int orc_primal_tableau(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                       int32_t* trace, int* n_pivots)
{
    int iter = 1;
    int np = 0;
    int status;
    for (;;) {
        if (iter > max_iter) { status = ORC_ITER_LIMIT; break; }      /* :95-96 */
        int entering = orc_choose_entering(T, R, C, eps);             /* :98 */
        if (entering == -1) { status = ORC_OPTIMAL; break; }          /* :99 */
        int leaving = orc_choose_leaving(T, R, C, entering, eps, eps);/* :101 */
        if (leaving == -1) { status = ORC_UNBOUNDED; break; }         /* :102-106 */
        orc_pivot(T, R, C, leaving, entering);                        /* :109 */
        basis[leaving] = entering;                                    /* :110 */
        if (trace) { trace[2 * np] = leaving; trace[2 * np + 1] = entering; }
        np++;
        iter++;
    }
    if (n_pivots) *n_pivots = np;
    return status;
}

void orc_forced_pivots(double* T, int R, int C, const int32_t* rows, const int32_t* cols,
                       int count, double thresh, int32_t* chosen)
{
    for (int k = 0; k < count; k++) {
        int r = rows[k];
        int q = -1;
        for (int s = 0; s < C; s++) {
            int j = cols[k] + s; if (j >= C) j -= C;
            if (fabs(T[(size_t)r * C + j]) >= thresh) { q = j; break; }
        }
        if (chosen) chosen[k] = q;
        if (q >= 0) orc_pivot(T, R, C, r, q);
    }
}

/* ExpandEqualitiesToInequalities (:161-177) + BuildTableau (:179-203) for an already
 * sense-normalised objective cmax[]. Returns malloc'ed T and basis; *pm = expanded row count. */
static double* build_tableau_primal(const orc_problem* p, const double* cmax, int* pm, int32_t** pbasis)
{
    int n = p->n;
    int m = 0;
    for (int i = 0; i < p->m; i++) m += (p->rel[i] == ORC_EQ) ? 2 : 1;
    int R = m + 1, C = n + m + 1;
    double* T = (double*)calloc((size_t)R * C, sizeof(double));
    int32_t* basis = (int32_t*)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    int row = 0;
    for (int i = 0; i < p->m; i++) {
        const double* a = p->A + (size_t)i * n;
        if (p->rel[i] == ORC_EQ) {
            /* +row <= b */
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = a[j];
            T[(size_t)row * C + n + row] = 1.0;
            T[(size_t)row * C + n + m] = p->b[i];
            row++;
            /* -row <= -b  (neg.A[j] *= -1; neg.B *= -1, :170-171) */
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = a[j] * -1;
            T[(size_t)row * C + n + row] = 1.0;
            T[(size_t)row * C + n + m] = p->b[i] * -1;
            row++;
        } else {
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = a[j];
            T[(size_t)row * C + n + row] = 1.0;          /* slack :191 */
            T[(size_t)row * C + n + m] = p->b[i];        /* :192 */
            row++;
        }
    }
    for (int j = 0; j < n; j++) T[(size_t)m * C + j] = -cmax[j];   /* :195 objective row is LAST */
    for (int i = 0; i < m; i++) basis[i] = n + i;                   /* :197 */
    *pm = m; *pbasis = basis;
    return T;
}

/* FinalizeReport, :130-159 (numbers only). */
static void finalize(orc_result* out, double* T, int R, int C, int32_t* basis, int n, int status)
{
    int m = R - 1;
    double* x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    for (int i = 0; i < m; i++)
        if (basis[i] < n) x[basis[i]] = T[(size_t)i * C + (C - 1)];
    out->status = status;
    out->has_solution = 1;
    out->z = T[(size_t)m * C + (C - 1)];   /* for Min inputs this is the NEGATED optimum, not flipped back */
    out->n = n; out->x = x; out->R = R; out->C = C; out->T = T; out->basis = basis;
}

/* PrimalSimplex.Solve, Models/PrimalSimplex.cs:57-127 */
int orc_primal_solve(const orc_problem* p, int max_iter, orc_result* out)
{
    memset(out, 0, sizeof(*out));
    int n = p->n;
    double* c = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    for (int j = 0; j < n; j++) c[j] = (p->sense == ORC_MIN) ? -p->c[j] : p->c[j];  /* :62-63 */
    for (int i = 0; i < p->m; i++) {                                               /* :66-77 */
        if (p->rel[i] == ORC_GE) { free(c); return ORC_E_GE_PRESENT; }
        if (p->b[i] < -1e-9)     { free(c); return ORC_E_NEG_RHS; }
    }
    int m; int32_t* basis;
    double* T = build_tableau_primal(p, c, &m, &basis);
    free(c);
    int R = m + 1, C = n + m + 1;
    int32_t* trace = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(max_iter > 0 ? max_iter : 1));
    int np = 0;
    int st = orc_primal_tableau(T, R, C, basis, 1e-9, max_iter, trace, &np);
    if (st == ORC_ITER_LIMIT) {   /* exception, :95-96: no result object */
        free(T); free(basis); free(trace);
        out->status = ORC_ITER_LIMIT; out->n_pivots = np;
        return ORC_ITER_LIMIT;
    }
    finalize(out, T, R, C, basis, n, st);
    out->n_pivots = np; out->trace = trace;
    return st;
}
