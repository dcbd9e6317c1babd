/* Courtesy multi-core CPU baseline ("all cores with OpenMP over rows of Pivot").
 * TEST / BENCH INFRASTRUCTURE ONLY -- see lpx_oracle.h.  The reference itself is single-threaded
 * (Program.cs:8); this variant only spreads the independent row updates of Pivot
 * (Models/PrimalSimplex.cs:250-256) over threads, so every element sees the same two roundings and the
 * result is bit-identical to orc_primal_tableau (tests/test_oracle_kats.py checks that). */
#include "lpx_oracle.h"
#include <omp.h>
#include <stddef.h>

static void pivot_mt(double* T, int R, int C, int r, int q)
{
    double* pr = T + (size_t)r * C;
    double piv = pr[q];
    for (int j = 0; j < C; j++) pr[j] /= piv;                       /* :248-249 */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < R; i++) {                                   /* :250-256 */
        if (i == r) continue;
        double* ti = T + (size_t)i * C;
        double factor = ti[q];
        for (int j = 0; j < C; j++) {
            double prod = factor * pr[j];
            ti[j] = ti[j] - prod;
        }
    }
}

int orc_primal_tableau_mt(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                          int32_t* trace, int* n_pivots, int threads)
{
    if (threads > 0) omp_set_num_threads(threads);
    omp_set_dynamic(0);
    int iter = 1, np = 0, status;
    for (;;) {
        if (iter > max_iter) { status = ORC_ITER_LIMIT; break; }
        int entering = orc_choose_entering(T, R, C, eps);
        if (entering < 0) { status = ORC_OPTIMAL; break; }
        int leaving = orc_choose_leaving(T, R, C, entering, eps, eps);
        if (leaving < 0) { status = ORC_UNBOUNDED; break; }
        pivot_mt(T, R, C, leaving, entering);
        basis[leaving] = entering;
        if (trace) { trace[2 * np] = leaving; trace[2 * np + 1] = entering; }
        np++;
        iter++;
    }
    if (n_pivots) *n_pivots = np;
    return status;
}
