/*
 * oracle/dual.c -- CPU restatement of Models/DualSimplex.cs (TEST INFRASTRUCTURE, see
 * lpx_oracle.h).
 *
 * Faithful mode (flags == 0) reproduces the two reference defects the B&B depends on:
 *   D1  PrepareForTableau turns `x >= b` (b > 0) into `-x <= -b` and then the "ensure b >= 0"
 *       step multiplies the row by -1 AGAIN (Models/DualSimplex.cs:141-153) -> `x <= b`.
 *   D2  FinalizeReport returns Report/Summary only (Models/DualSimplex.cs:310): no Solution,
 *       Tableau, Basis, VarNames; OptimalValue stays 0.
 * Repaired mode (ORC_DUAL_REPAIRED) is OUR definition ("B&B parallelism
 * vs reference defects"), built from the reference's own loops:
 *   bit0  skip the second sign flip (:148-153): negative right-hand sides reach the dual loop;
 *   bit1  return x / z / T / basis exactly as PrimalSimplex.FinalizeReport does (:130-159);
 *   bit2  ForceDualFeasibility guard 100 -> max_iter, and when the dual loop stops with every
 *         RHS >= -eps but the z-row still holds an entry < -eps (ForceDualFeasibility gave up on
 *         an unbounded column, :223), finish with the PrimalSimplex loop (:92-124) from that
 *         basis, which is primal feasible at that point.
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* ForceDualFeasibility, Models/DualSimplex.cs:195-228 */
static int force_dual_feasibility(double* T, int R, int C, int32_t* basis, double eps, double tol,
                                  int guard_max, int32_t* trace, int* np)
{
    int m = R - 1;
    (void)m;
    for (int guard = 0; guard < guard_max; guard++) {
        int entering = orc_choose_entering(T, R, C, eps);            /* :204-207, same scan as primal */
        if (entering == -1) return 1;                                /* :209 dual feasible */
        int leave = orc_choose_leaving(T, R, C, entering, eps, tol); /* :212-222, tol 1e-12 */
        if (leave == -1) return 0;                                   /* :223 */
        orc_pivot(T, R, C, leave, entering);                         /* :225 */
        basis[leave] = entering;                                     /* :226 */
        if (trace) { trace[2 * *np] = leave; trace[2 * *np + 1] = entering; }
        (*np)++;
    }
    return 0;
}

int orc_dual_tableau(double* T, int R, int C, int32_t* basis, double eps, double ratio_tol,
                     int fdf_guard, int max_iter, int cleanup,
                     int32_t* trace, int* n_pivots, int* n_fdf)
{
    int np = 0;
    force_dual_feasibility(T, R, C, basis, eps, ratio_tol, fdf_guard, trace, &np);   /* :24 */
    if (n_fdf) *n_fdf = np;
    int m = R - 1, nNoRhs = C - 1, rhsCol = C - 1;
    int iter = 1;
    int status;
    for (;;) {
        if (iter > max_iter) { status = ORC_ITER_LIMIT; break; }     /* :39 */
        /* leaving row: most negative RHS, first index (:46-55) */
        int leave = -1;
        double mostNeg = -eps;
        for (int i = 0; i < m; i++) {
            double v = T[(size_t)i * C + rhsCol];
            if (v < mostNeg) { mostNeg = v; leave = i; }
        }
        if (leave == -1) { status = ORC_OPTIMAL; break; }            /* :58-74 */
        /* entering column: min z_j / (-a) over a < -eps, hysteresis 1e-12 (:77-91) */
        int enter = -1;
        double bestRatio = INFINITY;
        const double* lr = T + (size_t)leave * C;
        const double* zr = T + (size_t)m * C;
        for (int j = 0; j < nNoRhs; j++) {
            double a = lr[j];
            if (a < -eps) {
                double ratio = zr[j] / (-a);
                if (ratio < bestRatio - ratio_tol) { bestRatio = ratio; enter = j; }
            }
        }
        if (enter == -1) { status = ORC_INFEASIBLE; break; }         /* :92-96 */
        orc_pivot(T, R, C, leave, enter);                            /* :99 */
        basis[leave] = enter;                                        /* :100 */
        if (trace) { trace[2 * np] = leave; trace[2 * np + 1] = enter; }
        np++;
        iter++;
    }
    if (cleanup && status == ORC_OPTIMAL && orc_choose_entering(T, R, C, eps) != -1) {
        int extra = 0;
        int left = max_iter - (iter - 1);
        status = orc_primal_tableau(T, R, C, basis, eps, left, trace ? trace + 2 * np : NULL, &extra);
        np += extra;
    }
    if (n_pivots) *n_pivots = np;
    return status;
}

/* DualSimplex.Solve, Models/DualSimplex.cs:15-114 */
int orc_dual_solve(const orc_problem* p, int flags, int max_iter, orc_result* out)
{
    memset(out, 0, sizeof(*out));
    const double Eps = 1e-9;                                          /* :13 */
    int n = p->n;
    /* PrepareForTableau, :117-158 */
    int m = 0;
    for (int i = 0; i < p->m; i++) m += (p->rel[i] == ORC_EQ) ? 2 : 1;
    int R = m + 1, C = n + m + 1;
    double* T = (double*)calloc((size_t)R * C, sizeof(double));
    int32_t* basis = (int32_t*)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    double* rowbuf = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    int row = 0;
    for (int i = 0; i < p->m; i++) {
        const double* a = p->A + (size_t)i * n;
        if (p->rel[i] == ORC_EQ) {                                    /* :129-137 */
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = a[j];
            T[(size_t)row * C + n + row] = 1.0;
            T[(size_t)row * C + n + m] = p->b[i];
            row++;
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = a[j] * -1;
            T[(size_t)row * C + n + row] = 1.0;
            T[(size_t)row * C + n + m] = -p->b[i];
            row++;
        } else {
            double B = p->b[i];
            for (int j = 0; j < n; j++) rowbuf[j] = a[j];
            if (p->rel[i] == ORC_GE) {                                /* :141-147 */
                for (int j = 0; j < n; j++) rowbuf[j] *= -1;
                B *= -1;
            }
            if (!(flags & ORC_DUAL_FIX_D1) && B < -Eps) {             /* :148-153  (D1) */
                for (int j = 0; j < n; j++) rowbuf[j] *= -1;
                B *= -1;
            }
            for (int j = 0; j < n; j++) T[(size_t)row * C + j] = rowbuf[j];
            T[(size_t)row * C + n + row] = 1.0;                       /* :173 */
            T[(size_t)row * C + n + m] = B;                           /* :174 */
            row++;
        }
    }
    free(rowbuf);
    for (int j = 0; j < n; j++) {                                     /* :122-123, :178 */
        double cj = (p->sense == ORC_MIN) ? -p->c[j] : p->c[j];
        T[(size_t)m * C + j] = -cj;
    }
    for (int i = 0; i < m; i++) basis[i] = n + i;                     /* :181 */

    int sound = (flags & ORC_DUAL_SOUND) != 0;
    int guard = sound ? max_iter : 100;                               /* :202 */
    size_t cap = (size_t)max_iter * (sound ? 3 : 1) + 128;
    int32_t* trace = (int32_t*)malloc(sizeof(int32_t) * 2 * cap);
    int np = 0, nfdf = 0;
    int st = orc_dual_tableau(T, R, C, basis, Eps, 1e-12, guard, max_iter, sound, trace, &np, &nfdf);
    out->n_pivots = np; out->n_fdf_pivots = nfdf;
    if (st == ORC_ITER_LIMIT) {                                       /* exception, :39 */
        free(T); free(basis); free(trace);
        out->status = ORC_ITER_LIMIT;
        return ORC_ITER_LIMIT;
    }
    out->status = st;
    out->trace = trace;
    if (flags & ORC_DUAL_FIX_D2) {
        double* x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
        for (int i = 0; i < m; i++)
            if (basis[i] < n) x[basis[i]] = T[(size_t)i * C + (C - 1)];   /* :290-297 */
        out->has_solution = 1;
        out->z = T[(size_t)m * C + (C - 1)];                            /* :299 */
        out->n = n; out->x = x; out->R = R; out->C = C; out->T = T; out->basis = basis;
    } else {
        /* D2: text only (:310).  OptimalValue defaults to 0, arrays null. */
        out->has_solution = 0;
        out->z = 0.0;
        out->n = n; out->R = R; out->C = C;
        free(T); free(basis);
    }
    return st;
}
