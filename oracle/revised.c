/*
 * oracle/revised.c -- CPU restatement of Models/RevisedPrimalSimplex.cs (TEST
 * INFRASTRUCTURE, see lpx_oracle.h).  As in the reference, B^-1 is recomputed by a full
 * Gauss-Jordan inversion every iteration (:128); summation orders of MultiplyRow / Multiply /
 * Dot follow the C# loops.
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

void orc_revised_result_free(orc_revised_result* r)
{
    if (!r) return;
    free(r->x); free(r->Bidx); free(r->Nidx); free(r->xB); free(r->trace);
    memset(r, 0, sizeof(*r));
}

/* Invert, Models/RevisedPrimalSimplex.cs:402-456: Gauss-Jordan on [M | I] with partial pivoting
 * (first maximum of |a| on ties, :419-425), singular if |pivot| < 1e-9 (:426). */
int orc_invert(const double* M, int n, double* inv)
{
    const double Eps = 1e-9;
    int w = 2 * n;
    double* A = (double*)calloc((size_t)n * w, sizeof(double));
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) A[(size_t)i * w + j] = M[(size_t)i * n + j];
        A[(size_t)i * w + n + i] = 1.0;
    }
    for (int col = 0; col < n; col++) {
        int pivotRow = col;
        double best = fabs(A[(size_t)pivotRow * w + col]);
        for (int r = col + 1; r < n; r++) {
            double v = fabs(A[(size_t)r * w + col]);
            if (v > best) { best = v; pivotRow = r; }
        }
        if (fabs(A[(size_t)pivotRow * w + col]) < Eps) { free(A); return ORC_E_SINGULAR; }
        if (pivotRow != col)
            for (int j = 0; j < w; j++) {
                double tmp = A[(size_t)col * w + j];
                A[(size_t)col * w + j] = A[(size_t)pivotRow * w + j];
                A[(size_t)pivotRow * w + j] = tmp;
            }
        double piv = A[(size_t)col * w + col];
        for (int j = 0; j < w; j++) A[(size_t)col * w + j] /= piv;
        for (int r = 0; r < n; r++) {
            if (r == col) continue;
            double factor = A[(size_t)r * w + col];
            double* ar = A + (size_t)r * w;
            const double* ac = A + (size_t)col * w;
            for (int j = 0; j < w; j++) { double prod = factor * ac[j]; ar[j] = ar[j] - prod; }
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) inv[(size_t)i * n + j] = A[(size_t)i * w + n + j];
    free(A);
    return 0;
}

/* Multiply(double[,] A, double[] x), :325-336 */
static void mat_vec(const double* A, int r, int c, const double* x, double* y)
{
    for (int i = 0; i < r; i++) {
        double s = 0;
        for (int j = 0; j < c; j++) s += A[(size_t)i * c + j] * x[j];
        y[i] = s;
    }
}

/* RevisedPrimalSimplex.Solve, :17-145 */
int orc_revised_solve(const orc_problem* p, int max_iter, orc_revised_result* out)
{
    const double Eps = 1e-9;                                          /* :14 */
    memset(out, 0, sizeof(*out));
    for (int i = 0; i < p->m; i++)                                    /* :19-21 */
        if (!(p->rel[i] == ORC_LE && p->b[i] >= -1e-9)) return ORC_E_REVISED_PRECOND;
    int m = p->m, n = p->n, Ntot = n + m;
    /* Standardize (:148-186): Max -> negate C (the solver MINIMISES c); the GE/EQ/negative-RHS
     * branches are dead code behind the :19 precondition. */
    double* A = (double*)calloc((size_t)m * Ntot, sizeof(double));
    double* c = (double*)calloc(Ntot, sizeof(double));
    double* b = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    for (int i = 0; i < m; i++) {                                     /* :36-42 */
        for (int j = 0; j < n; j++) A[(size_t)i * Ntot + j] = p->A[(size_t)i * n + j];
        A[(size_t)i * Ntot + n + i] = 1.0;
        b[i] = p->b[i];
    }
    for (int j = 0; j < n; j++) c[j] = (p->sense == ORC_MAX) ? -p->c[j] : p->c[j];  /* :43, :153-154 */
    int32_t* Bidx = (int32_t*)malloc(sizeof(int32_t) * (m > 0 ? m : 1));
    int32_t* Nidx = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    for (int i = 0; i < m; i++) Bidx[i] = n + i;                      /* :47 */
    for (int j = 0; j < n; j++) Nidx[j] = j;                          /* :48 */
    int nN = n;

    double* Bm = (double*)malloc(sizeof(double) * (size_t)m * m);
    double* Binv = (double*)malloc(sizeof(double) * (size_t)m * m);
    double* xB = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    double* cB = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    double* piT = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    double* rN = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    double* d = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    double* aq = (double*)malloc(sizeof(double) * (m > 0 ? m : 1));
    int32_t* trace = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(max_iter > 0 ? max_iter : 1));
    int status = ORC_ITER_LIMIT;
    int it_done = 0;
    int rc = 0;

#define GATHER_B() do { for (int i_ = 0; i_ < m; i_++) for (int j_ = 0; j_ < m; j_++) \
        Bm[(size_t)i_ * m + j_] = A[(size_t)i_ * Ntot + Bidx[j_]]; } while (0)

    GATHER_B();                                                       /* :58 */
    rc = orc_invert(Bm, m, Binv);
    if (rc) { status = ORC_E_SINGULAR; goto done; }
    mat_vec(Binv, m, m, b, xB);                                       /* :59 */
    for (int i = 0; i < m; i++) cB[i] = c[Bidx[i]];                   /* :60 */

    for (int iter = 1; iter <= max_iter; iter++) {                    /* :66 */
        /* piT = MultiplyRow(cB, Binv), :71 / :365-378 (i ascending inner loop) */
        for (int j = 0; j < m; j++) {
            double s = 0;
            for (int i = 0; i < m; i++) s += cB[i] * Binv[(size_t)i * m + j];
            piT[j] = s;
        }
        /* rN = cN - MultiplyRow(piT, Nmat), :69-72 */
        for (int j = 0; j < nN; j++) {
            int col = Nidx[j];
            double s = 0;
            for (int i = 0; i < m; i++) s += piT[i] * A[(size_t)i * Ntot + col];
            rN[j] = c[col] - s;
        }
        int enteringPos = -1;                                         /* :76-83 */
        double minRC = -Eps;
        for (int j = 0; j < nN; j++)
            if (rN[j] < minRC) { minRC = rN[j]; enteringPos = j; }
        if (enteringPos == -1) { status = ORC_OPTIMAL; break; }       /* :84-90 */
        int entering = Nidx[enteringPos];                             /* :92 */
        for (int i = 0; i < m; i++) aq[i] = A[(size_t)i * Ntot + entering];   /* :95 */
        mat_vec(Binv, m, m, aq, d);                                   /* :96 */
        int leaveRow = -1;                                            /* :99-112 */
        double bestTheta = INFINITY;
        for (int i = 0; i < m; i++) {
            if (d[i] > Eps) {
                double theta = xB[i] / d[i];
                if (theta < bestTheta - 1e-12) { bestTheta = theta; leaveRow = i; }
            }
        }
        if (leaveRow == -1) { status = ORC_UNBOUNDED; break; }        /* :113-118 */
        int leaving = Bidx[leaveRow];                                 /* :121-124 */
        Bidx[leaveRow] = entering;
        for (int j = enteringPos; j + 1 < nN; j++) Nidx[j] = Nidx[j + 1];   /* RemoveAt */
        Nidx[nN - 1] = leaving;                                             /* Add */
        trace[2 * it_done] = leaveRow; trace[2 * it_done + 1] = entering;
        it_done++;
        GATHER_B();                                                   /* :128 */
        rc = orc_invert(Bm, m, Binv);
        if (rc) { status = ORC_E_SINGULAR; break; }
        for (int i = 0; i < m; i++) cB[i] = c[Bidx[i]];               /* :131 */
        mat_vec(Binv, m, m, b, xB);                                   /* :132 */
    }
done:
    out->status = status;
    out->n = n; out->m = m;
    out->n_iters = it_done;
    out->trace = trace;
    out->Bidx = Bidx; out->Nidx = Nidx; out->xB = xB;
    out->x = (double*)calloc(n > 0 ? n : 1, sizeof(double));
    for (int i = 0; i < m; i++) if (Bidx[i] < n) out->x[Bidx[i]] = xB[i];     /* :270-274 */
    { double z = 0; for (int i = 0; i < m; i++) z += cB[i] * xB[i]; out->z_internal = z; }  /* :133 */
    { double z = 0; for (int j = 0; j < n; j++) z += p->c[j] * out->x[j]; out->z_original = z; } /* :287-289 */
    free(A); free(c); free(b); free(Bm); free(Binv); free(cB); free(piT); free(rN); free(d); free(aq);
    return status;
}
