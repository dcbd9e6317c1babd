// lpx_revised.hip -- revised primal simplex on gfx950: pricing / FTRAN as wave-per-vector GEMVs over
// HBM-resident A^T and B^-1, product-form update through the tableau path's lpx_update kernel.
//
// Device data (all FP64, row-major, leading dimensions padded to 16 doubles):
//   AT [n x ldat]   A transposed: column j of A is a contiguous row -> pricing is a row dot, and
//                   the entering column needs no strided gather.
//   W  [(m+1) x ldw] = [[B^-1, x_B], [pi, z]]  with pi = c_B B^-1, z = c_B x_B.  A pivot on row r with
//                   column vector f = [d; pi.a_q - c_q] (d = B^-1 a_q) is the Gauss-Jordan update of
//                   Models/PrimalSimplex.cs:245-257 applied to W -> reuses lpx_update unchanged,
//                   including its contiguous copy of the last column (x_B) for the ratio test.
//   key[n+m]        order key of nonbasic columns (-1 = basic).  Nidx.RemoveAt + Nidx.Add(leaving)
//                   (Models/RevisedPrimalSimplex.cs:123-124) preserves relative order, so "first in
//                   list order" == smallest key with keys handed out increasingly.
#include "lpx_block.h"
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <utility>

namespace lpx {

struct RvParams {
    int m, n, ldat, ldw;
    const double* AT; const double* c;   // c[n] (structural costs; slack costs are 0)
    double* W; double* prow; double* fac; double* rhsbuf; double* rc; double* aq;
    int32_t* Bidx; int32_t* key; int32_t* trace; int trace_cap;
    DevState* st;
    double eps, tol; int max_iter;
    double* ws; int rcap;
    double* part_v; int32_t* part_k; int32_t* part_c; int nblk;     // per-workgroup entering candidates of rv_price
};

static constexpr int RV_NT = 1024;
static constexpr int RV_NW = RV_NT / 64;

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// dot of two length-m vectors by one wave: 16-byte loads, four independent 1-KiB chunks in flight per
// operand and four accumulator pairs (a fixed, deterministic summation order; the revised path is compared
// on pivots and 1e-9 objective, not bitwise).  Both vectors are padded to a multiple of 16 doubles.
__device__ __forceinline__ double wave_dot(const double* __restrict__ x, const double* __restrict__ y, int m, int lane)
{
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, s4 = 0.0, s5 = 0.0, s6 = 0.0, s7 = 0.0;
    const int mp = m & ~1;
    int k = lane * 2;
    for (; k + 384 < mp; k += 512) {
        const double2 x0 = *reinterpret_cast<const double2*>(x + k),       y0 = *reinterpret_cast<const double2*>(y + k);
        const double2 x1 = *reinterpret_cast<const double2*>(x + k + 128), y1 = *reinterpret_cast<const double2*>(y + k + 128);
        const double2 x2 = *reinterpret_cast<const double2*>(x + k + 256), y2 = *reinterpret_cast<const double2*>(y + k + 256);
        const double2 x3 = *reinterpret_cast<const double2*>(x + k + 384), y3 = *reinterpret_cast<const double2*>(y + k + 384);
        s0 += x0.x * y0.x; s1 += x0.y * y0.y; s2 += x1.x * y1.x; s3 += x1.y * y1.y;
        s4 += x2.x * y2.x; s5 += x2.y * y2.y; s6 += x3.x * y3.x; s7 += x3.y * y3.y;
    }
    for (; k < mp; k += 128) {
        const double2 xv = *reinterpret_cast<const double2*>(x + k), yv = *reinterpret_cast<const double2*>(y + k);
        s0 += xv.x * yv.x; s1 += xv.y * yv.y;
    }
    if ((m & 1) && lane == 0) s0 += x[m - 1] * y[m - 1];
    return wave_sum(((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)));
}

// A (m x n, packed) -> AT (n x ldat): 32x32 tiles through LDS.
__global__ __launch_bounds__(256) void rv_transpose(const double* __restrict__ A, int m, int n,
                                                    double* __restrict__ AT, int ldat)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;       // bx: column of A, by: row of A
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int k = ty; k < 32; k += 8) {
        int i = by + k, j = bx + tx;
        tile[k][tx] = (i < m && j < n) ? A[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int j = bx + k, i = by + tx;
        if (j < n && i < m) AT[(size_t)j * ldat + i] = tile[tx][k];
    }
}

// rc[j] = c_j - pi . A[:,j]  for nonbasic structural j (+inf for basic ones).  One wave per column.
// MultiplyRow + Subtract, Models/RevisedPrimalSimplex.cs:71-72.
__global__ __launch_bounds__(256) void rv_price_dot(RvParams P)
{
    if (P.st->status != LPX_RUNNING) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= P.n) return;
    if (P.key[j] < 0) { if (lane == 0) P.rc[j] = __builtin_inf(); return; }
    const double* __restrict__ a = P.AT + (size_t)j * P.ldat;
    const double* __restrict__ pi = P.W + (size_t)P.m * P.ldw;
    const double s = wave_dot(a, pi, P.m, lane);
    if (lane == 0) P.rc[j] = P.c[j] - s;
}

struct Cand { double v; int key; int col; };
__device__ __forceinline__ Cand cand_pick(Cand a, Cand b)
{
    if (b.v < a.v || (b.v == a.v && b.key < a.key)) return b;
    return a;
}

// entering variable (Models/RevisedPrimalSimplex.cs:76-92) + dense copy of its column + the
// objective-row factor.  One workgroup.
__global__ __launch_bounds__(RV_NT) void rv_price_pick(RvParams P)
{
    __shared__ double s_v[RV_NW];
    __shared__ int s_k[RV_NW];
    __shared__ int s_c[RV_NW];
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (st->iter >= st->dual_iter) {                    // :66 / :144 -- the cap travels in the state record (DevState::dual_iter is
                                                        //   unused by this path), so the captured graph serves every segment of a run
        if (t == 0) { st->status = LPX_ITER_LIMIT; st->r = -1; }
        return;
    }
    const double* pi = P.W + (size_t)P.m * P.ldw;
    Cand best; best.v = -P.eps; best.key = INT_MAX; best.col = -1;
    for (int j = t; j < P.n; j += RV_NT) {
        const int k = P.key[j];
        if (k >= 0) { Cand c; c.v = P.rc[j]; c.key = k; c.col = j; if (c.v < -P.eps) best = cand_pick(best, c); }
    }
    for (int s = t; s < P.m; s += RV_NT) {
        const int k = P.key[P.n + s];
        const double rs = k >= 0 ? 0.0 - pi[s] : __builtin_inf();
        P.rc[P.n + s] = rs;                             // kept for lpx_revised_iteration_view (report text)
        if (k >= 0) { Cand c; c.v = rs; c.key = k; c.col = P.n + s; if (c.v < -P.eps) best = cand_pick(best, c); }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Cand o; o.v = __shfl_xor(best.v, d, 64); o.key = __shfl_xor(best.key, d, 64); o.col = __shfl_xor(best.col, d, 64);
        best = cand_pick(best, o);
    }
    if (lane == 0) { s_v[wave] = best.v; s_k[wave] = best.key; s_c[wave] = best.col; }
    __syncthreads();
    Cand w; w.v = s_v[lane & (RV_NW - 1)]; w.key = s_k[lane & (RV_NW - 1)]; w.col = s_c[lane & (RV_NW - 1)];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Cand o; o.v = __shfl_xor(w.v, d, 64); o.key = __shfl_xor(w.key, d, 64); o.col = __shfl_xor(w.col, d, 64);
        w = cand_pick(w, o);
    }
    const int q = w.col;
    if (q < 0) {                                        // :84-90
        if (t == 0) { st->status = LPX_OPTIMAL; st->r = -1; st->q = -1; }
        return;
    }
    for (int k = t; k < P.m; k += RV_NT)
        P.aq[k] = (q < P.n) ? P.AT[(size_t)q * P.ldat + k] : ((k == q - P.n) ? 1.0 : 0.0);
    if (t == 0) { st->q = q; P.fac[P.m] = -w.v; }       // pi.a_q - c_q = -(reduced cost)
}

// d = B^-1 a_q (Multiply, Models/RevisedPrimalSimplex.cs:96).  One wave per row of W.
__global__ __launch_bounds__(256) void rv_ftran(RvParams P)
{
    if (P.st->status != LPX_RUNNING) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= P.m) return;
    const double* __restrict__ w = P.W + (size_t)i * P.ldw;
    const double* __restrict__ a = P.aq;
    const double s = wave_dot(w, a, P.m, lane);
    if (lane == 0) P.fac[i] = s;
}

// Ratio test (Models/RevisedPrimalSimplex.cs:99-112, hysteresis 1e-12), normalisation of the pivot row
// of W, and the basis bookkeeping of :121-124.  One workgroup.
__global__ __launch_bounds__(SEL_NT) void rv_select(RvParams P)
{
    __shared__ int s_out;
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x;
    const int m = P.m;
    const int iter = st->iter;
    const int q = st->q;
    const double* d = P.fac;
    const int r = block_hysteresis_argmin(m, P.tol, RowRatio{P.fac, 1, P.rhsbuf, 1, P.eps}, &s_out);
    if (r < 0) {                                                    // :113-118
        if (t == 0) { st->status = LPX_UNBOUNDED; st->r = -1; }
        return;
    }
    const double piv = d[r];
    double* wrow = P.W + (size_t)r * P.ldw;
    const int C = m + 1;
    for (int j = t; j < C; j += SEL_NT) {
        const double p = wrow[j] / piv;
        wrow[j] = p;
        P.prow[j] = p;
    }
    __syncthreads();
    if (t == 0) {
        P.rhsbuf[r] = P.prow[m];
        const int leaving = P.Bidx[r];                              // :121-124
        P.Bidx[r] = q;
        P.key[q] = -1;
        P.key[leaving] = st->pad[0];
        st->pad[0] = st->pad[0] + 1;
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        st->iter = iter + 1;
        st->r = r; st->qn = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// Fused iteration (m <= RVF_MAXM): four launches instead of five and one pass less over W.
//
//   rv_price      r_N = c_N - pi N (MultiplyRow + Subtract, :71-72) with the entering candidate of :76-83 reduced on
//                 the way: a workgroup sweeps rows of A^T (non-temporal loads -- A^T is 268 MB at config 3 and would
//                 push W out of the Infinity Cache), each of its 8 waves owns a k-slice of pi in registers, and leaves
//                 ONE (reduced cost, order key, column) candidate per workgroup.
//   rv_pick       one workgroup: reduces the <= RVF_GRID candidates, copies the entering column (:95), iteration cap.
//   rv_upd_ftran  d = B^-1 a_q (:96) fused with the PREVIOUS iteration's rank-1 update of W: a row of W is read once,
//                 updated (same mul / sub per element as lpx_update), stored, and dotted with a_q in the same pass.  The
//                 update of iteration k is therefore applied lazily by iteration k+1 ("pending", DevState::pad[2..3]);
//                 lpx_revised_run and every accessor flush it (rv_flush) before W is looked at from outside.
//   rv_select2    one workgroup: ratio test (:99-112), pivot row normalised in place, the (pi, z) row updated EAGERLY
//                 (the next pricing needs it), x_B copy, basis bookkeeping (:121-124), pending := this pivot.
//
// Engine traffic per iteration: 8*m*n (A^T, read) + 16*m^2 (W, read + write) instead of 8*m*n + 24*m^2.
// ------------------------------------------------------------------------------------------------
static constexpr int RVF_NW = 8;                  // waves per workgroup
static constexpr int RVF_NT = RVF_NW * 64;
static constexpr int RVF_PER_MAX = 8;             // 1-KiB chunks of a row per wave held in registers (template: 1, 2, 4, 8)
static constexpr int RVF_MAXLD = RVF_NW * RVF_PER_MAX * 128; // m <= 8192
static constexpr int RVF_GRID = 2048;             // workgroups of rv_price (= candidates rv_pick reduces)
typedef double rv_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ Cand wave_cand_min(Cand c)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        Cand o; o.v = __shfl_xor(c.v, d, 64); o.key = __shfl_xor(c.key, d, 64); o.col = __shfl_xor(c.col, d, 64);
        c = cand_pick(c, o);
    }
    return c;
}

template <int RVF_PER>
__global__ __launch_bounds__(RVF_NT) void rv_price(RvParams P)
{
    __shared__ double s_part[2][RVF_NW];
    __shared__ double s_cv[RVF_NW];
    __shared__ int s_ck[RVF_NW], s_cc[RVF_NW];
    if (P.st->status != LPX_RUNNING) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m = P.m, n = P.n, ldat = P.ldat;
    const double* __restrict__ pi = P.W + (size_t)m * P.ldw;
    // chunk c of a row = doubles [128 c, 128 c + 128); wave w owns chunks w, w + NW, ...  (ldat is a multiple of 16 and A^T is
    // zero beyond column m, so the tail of the last chunk multiplies zeros as long as it stays inside the row)
    rv_d2 p[RVF_PER];
#pragma unroll
    for (int u = 0; u < RVF_PER; ++u) {
        const int k = (wave + u * RVF_NW) * 128 + lane * 2;
        p[u] = (k < m) ? *reinterpret_cast<const rv_d2*>(pi + k) : rv_d2{0.0, 0.0};
        if (k + 1 >= m && k < m) p[u].y = 0.0;                      // pi[m] is z, not a price
    }
    Cand best; best.v = -P.eps; best.key = INT_MAX; best.col = -1;
    for (int j0 = blockIdx.x * 2; j0 < n; j0 += gridDim.x * 2) {
        const int ja = j0, jb = min(j0 + 1, n - 1);
        const double* __restrict__ a0 = P.AT + (size_t)ja * ldat;
        const double* __restrict__ a1 = P.AT + (size_t)jb * ldat;
        rv_d2 x0[RVF_PER], x1[RVF_PER];
#pragma unroll
        for (int u = 0; u < RVF_PER; ++u) {
            const int k = (wave + u * RVF_NW) * 128 + lane * 2;
            if (k < ldat) {
                x0[u] = __builtin_nontemporal_load(reinterpret_cast<const rv_d2*>(a0 + k));
                x1[u] = __builtin_nontemporal_load(reinterpret_cast<const rv_d2*>(a1 + k));
            } else { x0[u] = rv_d2{0.0, 0.0}; x1[u] = rv_d2{0.0, 0.0}; }
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int u = 0; u < RVF_PER; ++u) {
            s0 += x0[u].x * p[u].x; s0 += x0[u].y * p[u].y;
            s1 += x1[u].x * p[u].x; s1 += x1[u].y * p[u].y;
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1);
        if (lane == 0) { s_part[0][wave] = s0; s_part[1][wave] = s1; }
        __syncthreads();
        if (threadIdx.x < 2 && j0 + (int)threadIdx.x < n) {
            const int j = j0 + threadIdx.x;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < RVF_NW; ++w) t += s_part[threadIdx.x][w];
            const int key = P.key[j];
            const double rcj = key >= 0 ? P.c[j] - t : __builtin_inf();
            P.rc[j] = rcj;
            if (key >= 0 && rcj < -P.eps) { Cand c; c.v = rcj; c.key = key; c.col = j; best = cand_pick(best, c); }
        }
        __syncthreads();
    }
    // slack columns: r_s = 0 - pi_s (their column of [A | I] is a unit vector); spread over the first workgroups
    const int s = blockIdx.x * RVF_NT + threadIdx.x;
    const bool has_slack = blockIdx.x * RVF_NT < m;                 // uniform per workgroup
    if (has_slack && s < m) {
        const int key = P.key[n + s];
        const double rs = key >= 0 ? 0.0 - pi[s] : __builtin_inf();
        P.rc[n + s] = rs;
        if (key >= 0 && rs < -P.eps) { Cand c; c.v = rs; c.key = key; c.col = n + s; best = cand_pick(best, c); }
    }
    // workgroup candidate: lanes 0,1 of wave 0 hold the structural ones, every lane possibly a slack one
    if (has_slack) {
        best = wave_cand_min(best);
        if (lane == 0) { s_cv[wave] = best.v; s_ck[wave] = best.key; s_cc[wave] = best.col; }
        __syncthreads();
        if (threadIdx.x == 0) {
            Cand w; w.v = s_cv[0]; w.key = s_ck[0]; w.col = s_cc[0];
#pragma unroll
            for (int k = 1; k < RVF_NW; ++k) { Cand o; o.v = s_cv[k]; o.key = s_ck[k]; o.col = s_cc[k]; w = cand_pick(w, o); }
            P.part_v[blockIdx.x] = w.v; P.part_k[blockIdx.x] = w.key; P.part_c[blockIdx.x] = w.col;
        }
    } else if (wave == 0) {
        Cand o; o.v = __shfl(best.v, 1, 64); o.key = __shfl(best.key, 1, 64); o.col = __shfl(best.col, 1, 64);
        best = cand_pick(best, o);
        if (lane == 0) { P.part_v[blockIdx.x] = best.v; P.part_k[blockIdx.x] = best.key; P.part_c[blockIdx.x] = best.col; }
    }
}

// entering variable from the workgroup candidates (:76-92), objective-row factor
__global__ __launch_bounds__(RV_NT) void rv_pick(RvParams P)
{
    __shared__ double s_v[RV_NW];
    __shared__ int s_k[RV_NW];
    __shared__ int s_c[RV_NW];
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (st->iter >= st->dual_iter) {                    // :66 / :144 -- the cap travels in the state record (DevState::dual_iter is
                                                        //   unused by this path), so the captured graph serves every segment of a run
        if (t == 0) { st->status = LPX_ITER_LIMIT; st->r = -1; }
        return;
    }
    Cand best; best.v = -P.eps; best.key = INT_MAX; best.col = -1;
    for (int b = t; b < P.nblk; b += RV_NT) {
        Cand c; c.v = P.part_v[b]; c.key = P.part_k[b]; c.col = P.part_c[b];
        if (c.col >= 0) best = cand_pick(best, c);
    }
    best = wave_cand_min(best);
    if (lane == 0) { s_v[wave] = best.v; s_k[wave] = best.key; s_c[wave] = best.col; }
    __syncthreads();
    Cand w; w.v = s_v[lane & (RV_NW - 1)]; w.key = s_k[lane & (RV_NW - 1)]; w.col = s_c[lane & (RV_NW - 1)];
    w = wave_cand_min(w);
    const int q = w.col;
    if (q < 0) {                                        // :84-90
        if (t == 0) { st->status = LPX_OPTIMAL; st->r = -1; st->q = -1; }
        return;
    }
    if (t == 0) { st->q = q; P.fac[P.m] = -w.v; }       // pi.a_q - c_q = -(reduced cost); a_q itself (:95) is read in place by rv_upd_ftran
}

// d = B^-1 a_q fused with the pending rank-1 update of the previous pivot (see the section comment).
// pad[2] = 1: rows i != pad[3] of W (i < m) still have to become  W[i,:] - fac[i] * prow[:]  (fac = previous d).
template <int RVF_PER>
__global__ __launch_bounds__(RVF_NT) void rv_upd_ftran(RvParams P)
{
    __shared__ double s_part[RVF_NW];
    const DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int m = P.m, ldw = P.ldw;
    const bool pending = st->pad[2] != 0;
    const int rp = st->pad[3];
    constexpr int cover = RVF_NW * RVF_PER * 128;       // >= m (launcher); columns [cover, ldw) are the tail below
    rv_d2 a[RVF_PER], p[RVF_PER];
    // a_q is read where it lies: row q of A^T, or the unit vector of a slack column (r03: rv_pick no longer copies it)
    const int q = st->q;
    const double* __restrict__ atq = (q < P.n) ? P.AT + (size_t)q * P.ldat : nullptr;
#pragma unroll
    for (int u = 0; u < RVF_PER; ++u) {
        const int k = (wave + u * RVF_NW) * 128 + lane * 2;
        a[u] = rv_d2{0.0, 0.0}; p[u] = rv_d2{0.0, 0.0};
        if (k < m) {                                                    // a_q is zero from m on: column m (x_B) stays out of the dot
            if (atq) { a[u].x = atq[k]; if (k + 1 < m) a[u].y = atq[k + 1]; }
            else { a[u].x = (k == q - P.n) ? 1.0 : 0.0; a[u].y = (k + 1 == q - P.n) ? 1.0 : 0.0; }
        }
        if (pending && k < ldw) p[u] = *reinterpret_cast<const rv_d2*>(P.prow + k);
    }
    for (int i = blockIdx.x; i < m; i += gridDim.x) {
        double* __restrict__ w = P.W + (size_t)i * ldw;
        const bool upd = pending && i != rp;
        const double f = upd ? P.fac[i] : 0.0;
        rv_d2 x[RVF_PER];
#pragma unroll
        for (int u = 0; u < RVF_PER; ++u) {
            const int k = (wave + u * RVF_NW) * 128 + lane * 2;
            x[u] = (k < ldw) ? *reinterpret_cast<const rv_d2*>(w + k) : rv_d2{0.0, 0.0};
        }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < RVF_PER; ++u) {
            const int k = (wave + u * RVF_NW) * 128 + lane * 2;
            if (upd && k < ldw) {
                x[u].x = x[u].x - f * p[u].x;            // mul, then sub -- the arithmetic of lpx_update
                x[u].y = x[u].y - f * p[u].y;
                *reinterpret_cast<rv_d2*>(w + k) = x[u];
            }
            s += x[u].x * a[u].x; s += x[u].y * a[u].y;
            if (k <= m && m < k + 2) P.rhsbuf[i] = (m == k) ? x[u].x : x[u].y;      // x_B[i] = W[i, m] (one lane of the workgroup)
        }
        // columns beyond the chunks (m a multiple of NW*128: the x_B column and its padding, <= 16 doubles); no part in the dot
        if (cover < ldw && (int)threadIdx.x < (ldw - cover) / 2) {
            const int k = cover + 2 * threadIdx.x;
            rv_d2 y = *reinterpret_cast<const rv_d2*>(w + k);
            if (upd) {
                const rv_d2 pt = *reinterpret_cast<const rv_d2*>(P.prow + k);
                y.x = y.x - f * pt.x; y.y = y.y - f * pt.y;
                *reinterpret_cast<rv_d2*>(w + k) = y;
            }
            if (k <= m && m < k + 2) P.rhsbuf[i] = (m == k) ? y.x : y.y;
        }
        s = wave_sum(s);
        if (lane == 0) s_part[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < RVF_NW; ++k) t += s_part[k];
            P.fac[i] = t;
        }
        __syncthreads();
    }
}

// Ratio test (:99-112, hysteresis 1e-12), pivot row normalised in place, (pi, z) row updated eagerly, bookkeeping (:121-124).
__global__ __launch_bounds__(SEL_NT) void rv_select2(RvParams P)
{
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x;
    const int m = P.m;
    const int iter = st->iter;
    const int q = st->q;
    const int r = block_hysteresis_segments<SEL_NW, RowRatio, true>(m, P.tol, RowRatio{P.fac, 1, P.rhsbuf, 1, P.eps});
    if (r < 0) {                                                    // :113-118
        if (t == 0) { st->status = LPX_UNBOUNDED; st->r = -1; }
        return;
    }
    const double piv = P.fac[r];
    const double fm = P.fac[m];
    double* wrow = P.W + (size_t)r * P.ldw;
    double* prw = P.W + (size_t)m * P.ldw;
    const int C = m + 1;
    __syncthreads();
    for (int j = t; j < C; j += SEL_NT) {
        const double p = wrow[j] / piv;                             // true division, as the tableau path
        wrow[j] = p;
        P.prow[j] = p;
        const double o = prw[j] - fm * p;                           // the (pi, z) row NOW: the next pricing reads it
        prw[j] = o;
        if (j == m) { P.rhsbuf[r] = p; P.rhsbuf[m] = o; }
    }
    for (int j = C + t; j < P.ldw; j += SEL_NT) P.prow[j] = 0.0;
    if (t == 0) {
        const int leaving = P.Bidx[r];                              // :121-124
        P.Bidx[r] = q;
        P.key[q] = -1;
        P.key[leaving] = st->pad[0];
        st->pad[0] = st->pad[0] + 1;
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        st->iter = iter + 1;
        st->r = r; st->qn = -1;
        st->pad[2] = 1; st->pad[3] = r;                             // pending: rows i != r, i < m
    }
}

// applies the pending update (if any) to W: the tableau path's tile, rows i < m, i != pad[3]
__global__ __launch_bounds__(256) void rv_flush(RvParams P, int ncw, int nunits)
{
    const DevState* st = P.st;
    if (st->pad[2] == 0) return;
    const int rp = st->pad[3];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= nunits) return;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= P.ldw) return;
    const rv_d2 p = *reinterpret_cast<const rv_d2*>(P.prow + col);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = rb * 8 + k;
        if (i < P.m && i != rp) {
            double* w = P.W + (size_t)i * P.ldw + col;
            rv_d2 v = *reinterpret_cast<const rv_d2*>(w);
            const double f = P.fac[i];
            v.x = v.x - f * p.x; v.y = v.y - f * p.y;
            *reinterpret_cast<rv_d2*>(w) = v;
        }
    }
}
__global__ void rv_clear_pending(DevState* st) { st->pad[2] = 0; }

// ------------------------------------------------------------------------------------------------
// K7': Invert (Models/RevisedPrimalSimplex.cs:402-456) on the device, bit for bit: Gauss-Jordan on the
// augmented [M | I] (n x 2n) with partial pivoting -- first maximum of |a| on ties (:419-425), singular if
// |pivot| < 1e-9 (:426), row swap (:428-434), scaling by true division (:437-438), elimination of every
// other row over all 2n columns (:441-446).  One step = inv_select (one workgroup: pivot search, swap,
// scale, column snapshot) + the tableau path's lpx_update on the n x 2n matrix (HBM-bound, 32 n^2 bytes).
// Used for the periodic refactorisation of the revised path and exposed as lpx_invert.
// ------------------------------------------------------------------------------------------------
struct InvParams {
    double* A; int ld; int n;          // augmented matrix, n rows, 2n columns
    double* prow; double* pcol; DevState* st;
};
static constexpr int LPX_SINGULAR_STATUS = 5;

__global__ __launch_bounds__(SEL_NT) void inv_select(InvParams P)
{
    __shared__ double s_v[SEL_NW];
    __shared__ int s_i[SEL_NW];
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x;
    const int n = P.n, w = 2 * n;
    const size_t ld = (size_t)P.ld;
    const int col = st->iter;
    if (col >= n) { if (t == 0) { st->status = LPX_OPTIMAL; st->r = -1; } return; }
    // pivotRow = first row >= col with the largest |A[r,col]|  -> lexicographic min of (-|a|, r)
    MinIdx m; m.v = __builtin_inf(); m.i = INT_MAX;
    for (int r = col + t; r < n; r += SEL_NT) {
        const double v = -fabs(P.A[(size_t)r * ld + col]);
        if (v < m.v) { m.v = v; m.i = r; }
    }
    m = block_min_idx(m, s_v, s_i);
    const int prw = m.i;
    if (prw == INT_MAX || !(-m.v >= 1e-9)) {                       // Math.Abs(...) < Eps -> singular (:426)
        if (t == 0) { st->status = LPX_SINGULAR_STATUS; st->r = -1; }
        return;
    }
    double* rc = P.A + (size_t)col * ld;
    double* rp = P.A + (size_t)prw * ld;
    const double piv = rp[col];                                    // value that lands in A[col,col] after the swap
    __syncthreads();
    for (int j = t; j < w; j += SEL_NT) {                          // swap (:428-434) fused with scaling (:437-438)
        const double a = rc[j], b = rp[j];
        const double p = b / piv;
        rc[j] = p; P.prow[j] = p;
        if (prw != col) rp[j] = a;
    }
    __syncthreads();
    for (int i = t; i < n; i += SEL_NT) P.pcol[i] = (i == col) ? 0.0 : P.A[(size_t)i * ld + col];   // factors (:444)
    if (t == 0) { st->iter = col + 1; st->r = col; st->qn = -1; }
}

// [M | I] from a packed n x n matrix already on the device
__global__ __launch_bounds__(256) void inv_build(const double* __restrict__ M, int n, int ldm, double* __restrict__ A, int ld)
{
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= ld) return;
    double v = 0.0;
    if (j < n) v = M[(size_t)i * ldm + j];
    else if (j == n + i) v = 1.0;
    A[(size_t)i * ld + j] = v;
}

// basis matrix B = [A | I][:, Bidx] gathered from AT (GetSubmatrix, :58,:128)
__global__ __launch_bounds__(256) void rv_gather_basis(RvParams P, double* __restrict__ M, int ldm)
{
    const int i = blockIdx.x * 256 + threadIdx.x, jb = blockIdx.y;     // M[i, jb] = column Bidx[jb], row i
    if (i >= P.m) return;
    const int col = P.Bidx[jb];
    M[(size_t)i * ldm + jb] = (col < P.n) ? P.AT[(size_t)col * P.ldat + i] : ((i == col - P.n) ? 1.0 : 0.0);
}

// W <- [[inv, inv*b], [cB*inv, cB*xB]] after a refactorisation (Multiply :59/:132, MultiplyRow :71, Dot :61/:133)
__global__ __launch_bounds__(256) void rv_refill_rows(RvParams P, const double* __restrict__ Ainv, int ldi, const double* __restrict__ b)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= P.m) return;
    const double* src = Ainv + (size_t)i * ldi + P.m;                 // right half of the augmented matrix
    double* dst = P.W + (size_t)i * P.ldw;
    double s = 0.0;
    for (int k = lane; k < P.m; k += 64) { const double v = src[k]; dst[k] = v; s += v * b[k]; }
    s = wave_sum(s);
    if (lane == 0) { dst[P.m] = s; P.rhsbuf[i] = s; }
}
__global__ __launch_bounds__(256) void rv_refill_pi(RvParams P, const double* __restrict__ call)
{
    // pi_j = sum_i cB_i * W[i,j]  (one lane per column, rows ascending as MultiplyRow does), z = cB . xB
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > P.m) return;
    double s = 0.0;
    for (int i = 0; i < P.m; ++i) {
        const int col = P.Bidx[i];
        const double cb = col < P.n ? call[col] : 0.0;
        s += cb * P.W[(size_t)i * P.ldw + j];
    }
    P.W[(size_t)P.m * P.ldw + j] = s;
    if (j == P.m) P.rhsbuf[P.m] = s;
}

// ---- drift control -------------------------------------------------------------------------------------------
// Cheap residual of the maintained inverse: rho = max_i |(B x_B)_i - b_i| / (1 + max_i |b_i|) with x_B the last column of W.
// B x_B = sum_k x_B[k] * column Bidx[k] of [A | I]; columns are rows of A^T, so thread i of a workgroup reads A^T[col][i]
// coalesced; the k range is split over blockIdx.y and the slices meet in a [KS][m] buffer (fixed order: deterministic).
static constexpr int RVR_KS = 32;
// probe vector of the second check (below): fixed, of order one, no zeros
__device__ __forceinline__ double rv_probe_v(int k) { return 0.75 + (double)((unsigned)k * 2654435761u >> 22) * (1.0 / 2048.0); }
// PROBE = false: y = B x_B;  true: y = B v with the probe vector v
template <bool PROBE>
__global__ __launch_bounds__(256) void rv_resid_partial(RvParams P, double* __restrict__ part)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per = (P.m + RVR_KS - 1) / RVR_KS;
    const int k0 = blockIdx.y * per, k1 = min(P.m, k0 + per);
    double acc = 0.0;
    for (int k = k0; k < k1; ++k) {
        const int col = P.Bidx[k];
        const double xk = PROBE ? rv_probe_v(k) : P.W[(size_t)k * P.ldw + P.m];
        if (i < P.m) acc += (col < P.n) ? P.AT[(size_t)col * P.ldat + i] * xk : ((i == col - P.n) ? xk : 0.0);
    }
    if (i < P.m) part[(size_t)blockIdx.y * P.m + i] = acc;
}
__global__ __launch_bounds__(1024) void rv_resid_final(RvParams P, const double* __restrict__ part, const double* __restrict__ b, double* out)
{
    __shared__ double s_a[16], s_b[16];
    double e = 0.0, bm = 0.0;
    for (int i = threadIdx.x; i < P.m; i += 1024) {
        double y = 0.0;
        for (int k = 0; k < RVR_KS; ++k) y += part[(size_t)k * P.m + i];
        e = fmax(e, fabs(y - b[i])); bm = fmax(bm, fabs(b[i]));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { e = fmax(e, __shfl_xor(e, d, 64)); bm = fmax(bm, __shfl_xor(bm, d, 64)); }
    if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = e; s_b[threadIdx.x >> 6] = bm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) { e = fmax(e, s_a[k]); bm = fmax(bm, s_b[k]); }
        out[0] = e / (1.0 + bm); out[1] = e;
    }
}

// Second probe (advisor finding r02: the residual above sees B^-1 only through b): w = B^-1 (B v) for a fixed probe vector v must
// give v back; rho2 = max_i |w_i - v_i| / (1 + max_i |v_i|) sees an error of the maintained inverse in a direction b does not
// excite.  y = B v comes from rv_resid_partial<true> (slices summed here, fixed order), w_i = W[i, 0..m) . y is one wave per row.
__global__ __launch_bounds__(256) void rv_probe_sum(RvParams P, const double* __restrict__ part, double* __restrict__ y)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.m) return;
    double s = 0.0;
    for (int k = 0; k < RVR_KS; ++k) s += part[(size_t)k * P.m + i];
    y[i] = s;
}
__global__ __launch_bounds__(256) void rv_probe_rows(RvParams P, const double* __restrict__ y, double* __restrict__ err)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= P.m) return;
    const double s = wave_dot(P.W + (size_t)i * P.ldw, y, P.m, lane);
    if (lane == 0) err[i] = fabs(s - rv_probe_v(i));
}
__global__ __launch_bounds__(1024) void rv_probe_final(RvParams P, const double* __restrict__ err, double* out)
{
    __shared__ double s_a[16];
    double e = 0.0;
    for (int i = threadIdx.x; i < P.m; i += 1024) e = fmax(e, err[i]);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) e = fmax(e, __shfl_xor(e, d, 64));
    if ((threadIdx.x & 63) == 0) s_a[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) { for (int k = 1; k < 16; ++k) e = fmax(e, s_a[k]); out[2] = e / (1.0 + 1.75); }     // |v_i| < 1.75
}

// fast refactorisation (Newton-Schulz on the matrix cores, lpx_mfma.hip): x_B = B^-1 b and (pi, z) = c_B [B^-1, x_B] from W in place
__global__ __launch_bounds__(256) void rv_refill_xb(RvParams P, const double* __restrict__ b)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= P.m) return;
    const double s = wave_dot(P.W + (size_t)i * P.ldw, b, P.m, lane);
    if (lane == 0) { P.W[(size_t)i * P.ldw + P.m] = s; P.rhsbuf[i] = s; }
}
static constexpr int RVP_RS = 16;
__global__ __launch_bounds__(256) void rv_refill_pi_partial(RvParams P, const double* __restrict__ call, double* __restrict__ part)
{
    // part[y][j] = sum over row slice y of cB_i * W[i][j], j <= m (column m gives z); 64 columns x 4 row groups per workgroup
    __shared__ double s_p[4][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c;
    const int per = (P.m + RVP_RS - 1) / RVP_RS;
    const int i0 = blockIdx.y * per, i1 = min(P.m, i0 + per);
    double acc = 0.0;
    if (j <= P.m)
        for (int i = i0 + g; i < i1; i += 4) {
            const int col = P.Bidx[i];
            const double cb = col < P.n ? call[col] : 0.0;
            acc += cb * P.W[(size_t)i * P.ldw + j];
        }
    s_p[g][c] = acc;
    __syncthreads();
    if (g == 0 && j <= P.m) part[(size_t)blockIdx.y * P.ldw + j] = (s_p[0][c] + s_p[1][c]) + (s_p[2][c] + s_p[3][c]);
}
__global__ __launch_bounds__(256) void rv_refill_pi_final(RvParams P, const double* __restrict__ part)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > P.m) return;
    double s = 0.0;
    for (int y = 0; y < RVP_RS; ++y) s += part[(size_t)y * P.ldw + j];
    P.W[(size_t)P.m * P.ldw + j] = s;
    if (j == P.m) P.rhsbuf[P.m] = s;
}

}  // namespace lpx

// ---------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ---------------------------------------------------------------------------------------------------
using namespace lpx;

namespace {

struct InvWork {
    int n = 0, ld = 0; double* A = nullptr; double* prow = nullptr; double* pcol = nullptr;
    DevState* st = nullptr; DevState* hst = nullptr; hipStream_t stream = nullptr;
    hipGraphExec_t gexec = nullptr; int g_batch = 0; std::string g_key; std::vector<hipEvent_t> events;
    ~InvWork() {
        if (stream) hipStreamSynchronize(stream);
        if (gexec) hipGraphExecDestroy(gexec);
        graph_cache_drop_owner(&gexec);
        hipFree(A); hipFree(prow); hipFree(pcol); hipFree(st);
        if (hst) hipHostFree(hst);
    }
};

int inv_alloc(InvWork& w, int n)
{
    w.n = n; w.ld = (2 * n + 15) / 16 * 16;
    LPX_HIP_TRY(hipMalloc((void**)&w.A, sizeof(double) * (size_t)n * w.ld));
    LPX_HIP_TRY(hipMalloc((void**)&w.prow, sizeof(double) * w.ld));
    LPX_HIP_TRY(hipMalloc((void**)&w.pcol, sizeof(double) * n));
    LPX_HIP_TRY(hipMalloc((void**)&w.st, sizeof(DevState)));
    LPX_HIP_TRY(hipHostMalloc((void**)&w.hst, sizeof(DevState)));
    if ((w.stream = borrow_stream()) == nullptr) { set_error("no stream"); return LPX_EDEVICE; }
    LPX_HIP_TRY(hipMemsetAsync(w.prow, 0, sizeof(double) * w.ld, w.stream));
    return 0;
}

// runs the n Gauss-Jordan steps on w.A (already holding [M | I]); returns 0 or LPX_E_SINGULAR
int inv_run(InvWork& w)
{
    InvParams ip; ip.A = w.A; ip.ld = w.ld; ip.n = w.n; ip.prow = w.prow; ip.pcol = w.pcol; ip.st = w.st;
    LoopCtx c;
    c.stream = w.stream; c.st = w.st; c.hst = w.hst; c.trace = nullptr; c.trace_cap = 0;
    c.events = &w.events; c.gexec = &w.gexec; c.g_batch = &w.g_batch; c.g_key = &w.g_key;
    c.key.assign(reinterpret_cast<const char*>(&ip), sizeof(ip));
    InvWork* pw = &w;
    c.enqueue_iter = [pw, ip](hipStream_t s, hipEvent_t e0, hipEvent_t e1) -> int {
        hipLaunchKernelGGL(inv_select, dim3(1), dim3(SEL_NT), 0, s, ip);
        LPX_HIP_TRY(hipGetLastError());
        LPX_HIP_TRY(launch_update(pw->A, pw->ld, pw->n, 2 * pw->n, nullptr, pw->prow, pw->pcol, pw->pcol, nullptr, pw->st, s, e0, e1));
        return 0;
    };
    c.launches_per_iter = 2; c.profile_maps = true;
    DevState init; std::memset(&init, 0, sizeof(init));
    init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = 2;
    lpx_run_opts o; lpx_default_opts(&o, 0); o.batch = 64;
    int st = run_device_loop(c, init, &o, (long long)w.n + 2, nullptr, nullptr, nullptr);
    if (st < 0) return st;
    if (st == LPX_SINGULAR_STATUS) { set_error("Singular basis encountered."); return LPX_E_SINGULAR; }
    if (st != LPX_OPTIMAL) { set_error("inversion did not finish"); return LPX_EDEVICE; }
    return 0;
}

}  // namespace

struct lpx_revised {
    int m = 0, n = 0, ldat = 0, ldw = 0;
    double *AT = nullptr, *c = nullptr, *W = nullptr, *prow = nullptr, *fac = nullptr, *rhsbuf = nullptr,
           *rc = nullptr, *aq = nullptr, *ws = nullptr;
    int32_t *Bidx = nullptr, *key = nullptr, *trace = nullptr;
    int trace_cap = 1 << 16;
    double* b = nullptr;            // right-hand sides (refactorisation)
    double* Mb = nullptr;           // m x m basis matrix scratch
    InvWork* inv = nullptr; int refactor_every = 0;
    double* part_v = nullptr; int32_t* part_k = nullptr; int32_t* part_c = nullptr;   // candidates of rv_price
    int refactor_mode = 0;          // 0 = exact Gauss-Jordan (the reference's Invert, bit for bit), 1 = Newton-Schulz on the matrix cores
    int drift_every = 256; double drift_tol = 1e-9;    // residual check of the maintained inverse (0 = off)
    double last_residual = -1.0, last_probe = -1.0; int refactors = 0, fast_steps = 0, fast_fallbacks = 0;
    double gemm_ms = 0.0; int gemm_calls = 0;       // HIP-event time of the matrix-core contractions of the last fast refactorisation
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *nsR = nullptr, *nsX = nullptr, *scratch = nullptr; unsigned long long* nsmax = nullptr; double* resid = nullptr;
    int ldp = 0;                    // leading dimension of the m x m scratch matrices (Mb, nsR)
    bool fused = false;             // four-launch iteration with the lazily applied W update (rows of W fit RVF_MAXLD)
    DevState* st = nullptr; DevState* hst = nullptr;
    hipStream_t stream = nullptr;
    hipGraphExec_t gexec = nullptr; int g_batch = 0; std::string g_key;
    std::vector<hipEvent_t> events;
};


static RvParams rv_params(lpx_revised* r, const lpx_run_opts* o)
{
    RvParams p; std::memset(&p, 0, sizeof(p));
    p.m = r->m; p.n = r->n; p.ldat = r->ldat; p.ldw = r->ldw;
    p.AT = r->AT; p.c = r->c; p.W = r->W; p.prow = r->prow; p.fac = r->fac; p.rhsbuf = r->rhsbuf;
    p.rc = r->rc; p.aq = r->aq; p.Bidx = r->Bidx; p.key = r->key; p.trace = r->trace; p.trace_cap = r->trace_cap;
    p.st = r->st; p.eps = o->eps; p.tol = o->ratio_tol; p.max_iter = 0;       // the iteration cap is DevState::dual_iter
    p.ws = r->ws; p.rcap = 0;
    p.part_v = r->part_v; p.part_k = r->part_k; p.part_c = r->part_c;
    p.nblk = std::min(RVF_GRID, std::max((r->n + 1) / 2, (r->m + RVF_NT - 1) / RVF_NT));
    return p;
}

template <int PER> static void rv_launch_fused(const RvParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(rv_price<PER>, dim3(p.nblk), dim3(RVF_NT), 0, s, p);
    hipLaunchKernelGGL(rv_pick, dim3(1), dim3(RV_NT), 0, s, p);
    hipLaunchKernelGGL(rv_upd_ftran<PER>, dim3(std::min(p.m, 4 * RVF_GRID)), dim3(RVF_NT), 0, s, p);
    hipLaunchKernelGGL(rv_select2, dim3(1), dim3(SEL_NT), 0, s, p);
}

// the same with every kernel bracketed by its own pair of HIP events (bound to the dispatch): ev[0..7] = {price, pick, upd_ftran, select2}
template <int PER> static void rv_launch_fused_profiled(const RvParams& p, hipStream_t s, hipEvent_t* ev)
{
    hipExtLaunchKernelGGL(rv_price<PER>, dim3(p.nblk), dim3(RVF_NT), 0, s, ev[0], ev[1], 0, p);
    hipExtLaunchKernelGGL(rv_pick, dim3(1), dim3(RV_NT), 0, s, ev[2], ev[3], 0, p);
    hipExtLaunchKernelGGL(rv_upd_ftran<PER>, dim3(std::min(p.m, 4 * RVF_GRID)), dim3(RVF_NT), 0, s, ev[4], ev[5], 0, p);
    hipExtLaunchKernelGGL(rv_select2, dim3(1), dim3(SEL_NT), 0, s, ev[6], ev[7], 0, p);
}

// applies the pending W update, if there is one (fused path); stream-ordered, then waits
static int rv_flush_pending(lpx_revised* r)
{
    if (!r->fused) return 0;
    lpx_run_opts od; lpx_default_opts(&od, 1);
    RvParams p = rv_params(r, &od);
    const int ncw = (p.ldw + 127) / 128, nunits = ncw * ((p.m + 7) / 8);
    hipLaunchKernelGGL(rv_flush, dim3((nunits + 3) / 4), dim3(256), 0, r->stream, p, ncw, nunits);
    hipLaunchKernelGGL(rv_clear_pending, dim3(1), dim3(1), 0, r->stream, r->st);
    LPX_HIP_TRY(hipGetLastError());
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    return 0;
}

static int rv_enqueue(lpx_revised* r, const RvParams& p, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    if (r->fused) {
        const int chunks = (p.m + 127) / 128, per = (chunks + RVF_NW - 1) / RVF_NW;     // chunks cover [0, m); see rv_upd_ftran's tail
        if (per <= 1) rv_launch_fused<1>(p, s); else if (per <= 2) rv_launch_fused<2>(p, s);
        else if (per <= 4) rv_launch_fused<4>(p, s); else rv_launch_fused<8>(p, s);
        LPX_HIP_TRY(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(rv_price_dot, dim3((p.n + 3) / 4), dim3(256), 0, s, p);
    hipLaunchKernelGGL(rv_price_pick, dim3(1), dim3(RV_NT), 0, s, p);
    hipLaunchKernelGGL(rv_ftran, dim3((p.m + 3) / 4), dim3(256), 0, s, p);
    hipLaunchKernelGGL(rv_select, dim3(1), dim3(SEL_NT), 0, s, p);
    LPX_HIP_TRY(hipGetLastError());
    LPX_HIP_TRY(launch_update(p.W, p.ldw, p.m + 1, p.m + 1, nullptr, p.prow, p.fac, p.fac, p.rhsbuf, p.st, s, e0, e1));
    return 0;
}

extern "C" {

void lpx_revised_destroy(lpx_revised* r)
{
    if (!r) return;
    if (r->stream) hipStreamSynchronize(r->stream);
    if (r->gexec) hipGraphExecDestroy(r->gexec);
    graph_cache_drop_owner(&r->gexec);
    for (hipEvent_t e : r->events) hipEventDestroy(e);
    if (r->ev0) hipEventDestroy(r->ev0);
    if (r->ev1) hipEventDestroy(r->ev1);
    hipFree(r->AT); hipFree(r->c); hipFree(r->W); hipFree(r->prow); hipFree(r->fac); hipFree(r->rhsbuf);
    hipFree(r->rc); hipFree(r->aq); hipFree(r->ws); hipFree(r->Bidx); hipFree(r->key); hipFree(r->trace);
    hipFree(r->nsR); hipFree(r->nsX); hipFree(r->scratch); hipFree(r->nsmax); hipFree(r->resid);
    hipFree(r->st); hipFree(r->b); hipFree(r->Mb); hipFree(r->part_v); hipFree(r->part_k); hipFree(r->part_c);
    delete r->inv;
    if (r->hst) hipHostFree(r->hst);
    delete r;
}

int lpx_revised_create(int m, int n, const double* A, const double* c, const double* b, lpx_revised** out)
{
    if (!out || m < 1 || n < 1 || !A || !c || !b) { set_error("lpx_revised_create: bad argument"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    lpx_revised* r = new lpx_revised();
    r->m = m; r->n = n;
    r->ldat = (m + 15) / 16 * 16;
    r->ldw = (m + 1 + 15) / 16 * 16;
    r->ldp = (m + 15) / 16 * 16;
    const size_t atb = sizeof(double) * (size_t)n * r->ldat;
    const size_t wb = sizeof(double) * (size_t)(m + 1) * r->ldw;
    double* Atmp = nullptr;
#define RALLOC(ptr, bytes)                                                              \
    do { hipError_t e_ = malloc_retry((void**)&(ptr), (bytes));                            \
         if (e_ != hipSuccess) { set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e_)); \
             hipFree(Atmp); lpx_revised_destroy(r); return e_ == hipErrorOutOfMemory ? LPX_ENOMEM : LPX_EDEVICE; } } while (0)
    RALLOC(r->AT, atb); RALLOC(r->c, sizeof(double) * n); RALLOC(r->W, wb);
    RALLOC(r->prow, sizeof(double) * r->ldw); RALLOC(r->fac, sizeof(double) * (m + 1));
    RALLOC(r->rhsbuf, sizeof(double) * (m + 1)); RALLOC(r->rc, sizeof(double) * (n + m));
    RALLOC(r->aq, sizeof(double) * r->ldw); RALLOC(r->ws, sizeof(double) * (m + 1));
    RALLOC(r->part_v, sizeof(double) * RVF_GRID); RALLOC(r->part_k, sizeof(int32_t) * RVF_GRID); RALLOC(r->part_c, sizeof(int32_t) * RVF_GRID);
    {   // LPX_REVISED_FUSED=0: the five-launch iteration with the eager W update (also the path of rows longer than RVF_MAXLD)
        static const bool fused_env = [] { const char* e = std::getenv("LPX_REVISED_FUSED"); return !(e && e[0] == '0'); }();
        r->fused = fused_env && m <= RVF_MAXLD;
    }
    RALLOC(r->Bidx, sizeof(int32_t) * m); RALLOC(r->key, sizeof(int32_t) * (n + m));
    RALLOC(r->trace, sizeof(int32_t) * 2 * r->trace_cap); RALLOC(r->st, sizeof(DevState));
    RALLOC(Atmp, sizeof(double) * (size_t)m * n);
    RALLOC(r->b, sizeof(double) * m);
#undef RALLOC
    if (hipHostMalloc((void**)&r->hst, sizeof(DevState)) != hipSuccess ||
        (r->stream = borrow_stream()) == nullptr) {
        set_error("pinned state / stream creation failed"); hipFree(Atmp); lpx_revised_destroy(r); return LPX_EDEVICE;
    }
    hipStream_t s = r->stream;
    hipMemsetAsync(r->st, 0, sizeof(DevState), s);
    hipMemsetAsync(r->AT, 0, atb, s);
    hipMemsetAsync(r->W, 0, wb, s);
    hipMemsetAsync(r->prow, 0, sizeof(double) * r->ldw, s);
    hipMemsetAsync(r->aq, 0, sizeof(double) * r->ldw, s);
    hipMemsetAsync(r->fac, 0, sizeof(double) * (m + 1), s);
    hipMemsetAsync(r->rhsbuf, 0, sizeof(double) * (m + 1), s);
    hipMemsetAsync(r->rc, 0, sizeof(double) * (n + m), s);          // nothing a fresh handle reads is left as hipMalloc found it
    hipMemsetAsync(r->ws, 0, sizeof(double) * (m + 1), s);
    hipMemsetAsync(r->trace, 0, sizeof(int32_t) * 2 * r->trace_cap, s);
    hipMemcpyAsync(Atmp, A, sizeof(double) * (size_t)m * n, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(r->c, c, sizeof(double) * n, hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(rv_transpose, dim3((n + 31) / 32, (m + 31) / 32), dim3(256), 0, s, Atmp, m, n, r->AT, r->ldat);
    // W = [[I, b], [0, 0]]  (Invert of the slack basis is exactly I, :58; xB = b, :59; pi = 0, z = 0)
    std::vector<double> one(m, 1.0);
    hipMemcpy2DAsync(r->W, sizeof(double) * (r->ldw + 1), one.data(), sizeof(double), sizeof(double), m, hipMemcpyHostToDevice, s);
    hipMemcpy2DAsync(r->W + m, sizeof(double) * r->ldw, b, sizeof(double), sizeof(double), m, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(r->rhsbuf, b, sizeof(double) * m, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(r->b, b, sizeof(double) * m, hipMemcpyHostToDevice, s);
    std::vector<int32_t> bidx(m), key(n + m);
    for (int i = 0; i < m; ++i) bidx[i] = n + i;                 // :47
    for (int j = 0; j < n; ++j) key[j] = j;                       // :48
    for (int i = 0; i < m; ++i) key[n + i] = -1;
    hipMemcpyAsync(r->Bidx, bidx.data(), sizeof(int32_t) * m, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(r->key, key.data(), sizeof(int32_t) * (n + m), hipMemcpyHostToDevice, s);
    hipError_t e = hipStreamSynchronize(s);
    hipFree(Atmp);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error(std::string("lpx_revised_create: upload failed: ") + hipGetErrorString(e));
        lpx_revised_destroy(r); return LPX_EDEVICE;
    }
    *out = r;
    return 0;
}

// basis matrix B = [A | I][:, Bidx] into r->Mb (m x ldp, zero padded)
static int rv_gather(lpx_revised* r, const RvParams& p, hipStream_t s)
{
    const int m = r->m;
    if (!r->Mb) {
        LPX_HIP_TRY(hipMalloc((void**)&r->Mb, sizeof(double) * (size_t)m * r->ldp));
        LPX_HIP_TRY(hipMemsetAsync(r->Mb, 0, sizeof(double) * (size_t)m * r->ldp, s));
    }
    hipLaunchKernelGGL(rv_gather_basis, dim3((m + 255) / 256, m), dim3(256), 0, s, p, r->Mb, r->ldp);
    LPX_HIP_TRY(hipGetLastError());
    return 0;
}

// exact refactorisation: the reference's Invert on the device, bit for bit (K7')
static int rv_refactor_exact(lpx_revised* r, const RvParams& p)
{
    const int m = r->m;
    // W is rebuilt from the basis: a rank-1 update still pending on the old W is dropped, not applied
    if (r->fused) { hipLaunchKernelGGL(rv_clear_pending, dim3(1), dim3(1), 0, r->stream, r->st); LPX_HIP_TRY(hipStreamSynchronize(r->stream)); }
    if (!r->inv) {
        r->inv = new InvWork();
        int rc = inv_alloc(*r->inv, m);
        if (rc) { delete r->inv; r->inv = nullptr; return rc; }
    }
    InvWork& w = *r->inv;
    { int rc = rv_gather(r, p, w.stream); if (rc) return rc; }
    hipLaunchKernelGGL(inv_build, dim3((w.ld + 255) / 256, m), dim3(256), 0, w.stream, (const double*)r->Mb, m, r->ldp, w.A, w.ld);
    LPX_HIP_TRY(hipGetLastError());
    int rc = inv_run(w);
    if (rc) return rc;
    hipLaunchKernelGGL(rv_refill_rows, dim3((m + 3) / 4), dim3(256), 0, w.stream, p, (const double*)w.A, w.ld, (const double*)r->b);
    hipLaunchKernelGGL(rv_refill_pi, dim3((m + 1 + 255) / 256), dim3(256), 0, w.stream, p, (const double*)r->c);
    LPX_HIP_TRY(hipGetLastError());
    LPX_HIP_TRY(hipStreamSynchronize(w.stream));
    return 0;
}

// Fast refactorisation: Newton-Schulz steps X <- X + X (I - B X) on the FP64 matrix cores (lpx_mfma.hip), starting from the
// maintained inverse.  Returns 1 when X is too far from B^-1 for the step to contract (the caller then runs the exact one).
static int rv_refactor_fast(lpx_revised* r, const RvParams& p)
{
    const int m = r->m, ldp = r->ldp, ldw = r->ldw;
    hipStream_t s = r->stream;
    { int rc = rv_flush_pending(r); if (rc) return rc; }             // the starting X is W as the last pivot left it
    if (!r->nsR) {
        LPX_HIP_TRY(hipMalloc((void**)&r->nsR, sizeof(double) * (size_t)m * ldp));
        LPX_HIP_TRY(hipMalloc((void**)&r->nsX, sizeof(double) * (size_t)m * ldw));
        LPX_HIP_TRY(hipMalloc((void**)&r->nsmax, sizeof(unsigned long long)));
        LPX_HIP_TRY(hipMalloc((void**)&r->scratch, sizeof(double) * (size_t)RVP_RS * ldw));
    }
    { int rc = rv_gather(r, p, s); if (rc) return rc; }
    if (!r->ev0) { LPX_HIP_TRY(hipEventCreate(&r->ev0)); LPX_HIP_TRY(hipEventCreate(&r->ev1)); }
    r->gemm_ms = 0.0; r->gemm_calls = 0;
    auto timed = [&](float& acc_ms) -> int { LPX_HIP_TRY(hipEventSynchronize(r->ev1)); float ms = 0.f; LPX_HIP_TRY(hipEventElapsedTime(&ms, r->ev0, r->ev1)); acc_ms += ms; return 0; };
    float gms = 0.f;
    for (int step = 0; step < 3; ++step) {
        LPX_HIP_TRY(hipMemsetAsync(r->nsmax, 0, sizeof(unsigned long long), s));
        LPX_HIP_TRY(hipEventRecord(r->ev0, s));
        LPX_HIP_TRY(launch_dgemm_mfma(r->Mb, ldp, r->W, ldw, r->nsR, ldp, nullptr, 0, m, m, m, 0, r->nsmax, s));      // R = I - B X
        LPX_HIP_TRY(hipEventRecord(r->ev1, s));
        unsigned long long bits = 0;
        LPX_HIP_TRY(hipMemcpyAsync(&bits, r->nsmax, sizeof(bits), hipMemcpyDeviceToHost, s));
        LPX_HIP_TRY(hipStreamSynchronize(s));
        double rmax; std::memcpy(&rmax, &bits, sizeof(rmax));
        { int rc = timed(gms); if (rc) return rc; } r->gemm_calls++; r->gemm_ms = gms;
        if (!(rmax * m < 0.1)) return 1;                             // not a contraction for sure: |R|_inf <= m * max|R_ij|
        if (step > 0 && rmax * m < 1e-13) break;                     // already at working precision
        LPX_HIP_TRY(hipEventRecord(r->ev0, s));
        LPX_HIP_TRY(launch_dgemm_mfma(r->W, ldw, r->nsR, ldp, r->nsX, ldw, r->W, ldw, m, m, m, 1, nullptr, s));        // X' = X + X R
        LPX_HIP_TRY(hipEventRecord(r->ev1, s));
        { int rc = timed(gms); if (rc) return rc; } r->gemm_calls++; r->gemm_ms = gms;
        LPX_HIP_TRY(hipMemcpy2DAsync(r->W, sizeof(double) * ldw, r->nsX, sizeof(double) * ldw, sizeof(double) * m, m, hipMemcpyDeviceToDevice, s));
        r->fast_steps++;
        if ((double)m * rmax * rmax * m < 1e-13) break;              // residual of X' is R^2: |R^2|_inf <= (m max|R_ij|)^2
    }
    hipLaunchKernelGGL(rv_refill_xb, dim3((m + 3) / 4), dim3(256), 0, s, p, (const double*)r->b);
    hipLaunchKernelGGL(rv_refill_pi_partial, dim3((m + 1 + 63) / 64, RVP_RS), dim3(256), 0, s, p, (const double*)r->c, r->scratch);
    hipLaunchKernelGGL(rv_refill_pi_final, dim3((m + 1 + 255) / 256), dim3(256), 0, s, p, (const double*)r->scratch);
    LPX_HIP_TRY(hipGetLastError());
    LPX_HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int lpx_revised_refactor(lpx_revised* r)
{
    if (!r) { set_error("lpx_revised_refactor: null handle"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    lpx_run_opts od; lpx_default_opts(&od, 1);
    RvParams p = rv_params(r, &od);
    r->refactors++;
    if (r->refactor_mode == 1) {
        const int rc = rv_refactor_fast(r, p);
        if (rc <= 0) return rc;
        r->fast_fallbacks++;                                         // X too far from B^-1: exact inversion instead
    }
    return rv_refactor_exact(r, p);
}

int lpx_revised_set_refactor_mode(lpx_revised* r, int mode)
{
    if (!r || mode < 0 || mode > 1) { set_error("lpx_revised_set_refactor_mode: mode is 0 (exact) or 1 (fast)"); return LPX_EINVAL; }
    r->refactor_mode = mode;
    return 0;
}

int lpx_revised_set_drift_policy(lpx_revised* r, int check_every, double tol)
{
    if (!r || check_every < 0 || !(tol >= 0.0)) { set_error("lpx_revised_set_drift_policy: bad argument"); return LPX_EINVAL; }
    r->drift_every = check_every; r->drift_tol = tol;
    return 0;
}

int lpx_revised_residual(lpx_revised* r, double* rel, double* abs_)
{
    if (!r) { set_error("lpx_revised_residual: null handle"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    { const int rc = rv_flush_pending(r); if (rc) return rc; }
    const int m = r->m;
    // layout of r->resid: [0] rho, [1] abs, [2] rho2, [3] pad, [4 ..) the [KS][m] slices, then y[m], then err[m]
    if (!r->resid) LPX_HIP_TRY(hipMalloc((void**)&r->resid, sizeof(double) * ((size_t)(RVR_KS + 2) * m + 4)));
    lpx_run_opts od; lpx_default_opts(&od, 1);
    RvParams p = rv_params(r, &od);
    double* part = r->resid + 4; double* y = part + (size_t)RVR_KS * m; double* err = y + m;
    hipLaunchKernelGGL(rv_resid_partial<false>, dim3((m + 255) / 256, RVR_KS), dim3(256), 0, r->stream, p, part);
    hipLaunchKernelGGL(rv_resid_final, dim3(1), dim3(1024), 0, r->stream, p, (const double*)part, (const double*)r->b, r->resid);
    hipLaunchKernelGGL(rv_resid_partial<true>, dim3((m + 255) / 256, RVR_KS), dim3(256), 0, r->stream, p, part);
    hipLaunchKernelGGL(rv_probe_sum, dim3((m + 255) / 256), dim3(256), 0, r->stream, p, (const double*)part, y);
    hipLaunchKernelGGL(rv_probe_rows, dim3((m + 3) / 4), dim3(256), 0, r->stream, p, (const double*)y, err);
    hipLaunchKernelGGL(rv_probe_final, dim3(1), dim3(1024), 0, r->stream, p, (const double*)err, r->resid);
    LPX_HIP_TRY(hipGetLastError());
    double out[3] = {0, 0, 0};
    LPX_HIP_TRY(hipMemcpyAsync(out, r->resid, sizeof(out), hipMemcpyDeviceToHost, r->stream));
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    r->last_residual = out[0] > out[2] ? out[0] : out[2];      // the larger of the two probes drives the drift policy
    r->last_probe = out[2];
    if (rel) *rel = r->last_residual;
    if (abs_) *abs_ = out[1];
    return 0;
}

int lpx_revised_refactor_stats(lpx_revised* r, int* refactors, int* fast_steps, int* fast_fallbacks, double* last_residual,
                               double* gemm_ms, int* gemm_calls)
{
    if (!r) return LPX_EINVAL;
    if (gemm_ms) *gemm_ms = r->gemm_ms;
    if (gemm_calls) *gemm_calls = r->gemm_calls;
    if (refactors) *refactors = r->refactors;
    if (fast_steps) *fast_steps = r->fast_steps;
    if (fast_fallbacks) *fast_fallbacks = r->fast_fallbacks;
    if (last_residual) *last_residual = r->last_residual;
    return 0;
}

int lpx_revised_set_refactor(lpx_revised* r, int every)
{
    if (!r || every < 0) return LPX_EINVAL;
    r->refactor_every = every;
    return 0;
}

static int revised_run_segment(lpx_revised* r, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st, int iter0);

int lpx_revised_run(lpx_revised* r, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    if (!r) { set_error("lpx_revised_run: null handle"); return LPX_EINVAL; }
    lpx_run_opts d; if (!o) { lpx_default_opts(&d, 1); o = &d; }
    // Segments: `refactor_every` iterations followed by an unconditional refactorisation (k = 1 is the reference's own
    // schedule, :128), or -- the default -- `drift_every` iterations followed by the residual check of the maintained
    // inverse and a refactorisation only when it has drifted past `drift_tol`.
    const int seg_len = r->refactor_every > 0 ? r->refactor_every : r->drift_every;
    if (seg_len <= 0 || seg_len >= o->max_iter) {
        const int status = revised_run_segment(r, o, cb, user, st, 0);
        const int rc = rv_flush_pending(r);             // W as the last pivot left it, before anyone looks at it
        return rc ? rc : status;
    }
    lpx_stats total; std::memset(&total, 0, sizeof(total));
    int done = 0, status = LPX_ITER_LIMIT;
    while (done < o->max_iter) {
        lpx_run_opts seg = *o;
        seg.max_iter = std::min(o->max_iter, done + seg_len);
        lpx_stats s1; std::memset(&s1, 0, sizeof(s1));
        status = revised_run_segment(r, &seg, cb, user, &s1, done);
        total.launches += s1.launches; total.loop_ms += s1.loop_ms; total.update_ms_sum += s1.update_ms_sum;
        total.update_launches += s1.update_launches; total.pivots = s1.pivots;
        if (status != LPX_ITER_LIMIT) break;
        done = (int)s1.pivots;
        if (done >= o->max_iter) break;
        const double t0 = now_ms();
        if (r->refactor_every > 0) {
            int rc = lpx_revised_refactor(r);           // exact mode drops the pending update: W is rebuilt from the basis
            if (rc) return rc;
        } else {
            double rel = 0.0;
            int rc = lpx_revised_residual(r, &rel, nullptr);
            if (rc) return rc;
            if (!(rel <= r->drift_tol)) { rc = lpx_revised_refactor(r); if (rc) return rc; }
        }
        total.loop_ms += now_ms() - t0;
    }
    if (st) { double h = st->h2d_ms, dd = st->d2h_ms; *st = total; st->h2d_ms = h; st->d2h_ms = dd; }
    { const int rc = rv_flush_pending(r); if (rc) return rc; }
    return status;
}

static int revised_run_segment(lpx_revised* r, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st, int iter0)
{
    RvParams p = rv_params(r, o);
    LoopCtx c;
    c.stream = r->stream; c.st = r->st; c.hst = r->hst; c.trace = r->trace; c.trace_cap = r->trace_cap;
    c.events = &r->events; c.gexec = &r->gexec; c.g_batch = &r->g_batch; c.g_key = &r->g_key;
    c.key.assign(reinterpret_cast<const char*>(&p), sizeof(p));
    c.enqueue_iter = [r, p](hipStream_t s, hipEvent_t e0, hipEvent_t e1) -> int { return rv_enqueue(r, p, s, e0, e1); };
    c.launches_per_iter = r->fused ? 4 : 5;
    c.profile_maps = true;
    DevState init; std::memset(&init, 0, sizeof(init));
    init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = 2;
    init.iter = iter0;             // continuing after a refactorisation keeps the iteration count and the trace
    init.dual_iter = o->max_iter;  // iteration cap of this segment, read by rv_pick / rv_price_pick
    // continue from the handle's current basis: the order-key counter lives in pad[0]
    LPX_HIP_TRY(hipMemcpy(r->hst, r->st, sizeof(DevState), hipMemcpyDeviceToHost));
    init.pad[0] = r->hst->pad[0] > 0 ? r->hst->pad[0] : r->n;
    return run_device_loop(c, init, o, (long long)o->max_iter + 2, cb, user, st);
}

// Per-kernel durations of the four-launch iteration ("report per-kernel achieved GB/s"): runs up to `iters` iterations
// from the handle's current basis eagerly, every kernel between its own HIP events; us[4] = mean microseconds of rv_price, rv_pick,
// rv_upd_ftran, rv_select2 over the iterations that completed a pivot.  The iterations are real ones (the basis moves on).
int lpx_revised_profile(lpx_revised* r, int iters, double* us, int* measured)
{
    if (!r || !us || iters < 1) { set_error("lpx_revised_profile: bad argument"); return LPX_EINVAL; }
    if (!r->fused) { set_error("lpx_revised_profile: only the four-launch iteration is instrumented"); return LPX_EINVAL; }
    if (iters > 512) iters = 512;
    lpx_run_opts o; lpx_default_opts(&o, 1);
    RvParams p = rv_params(r, &o);
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    LPX_HIP_TRY(hipMemcpy(r->hst, r->st, sizeof(DevState), hipMemcpyDeviceToHost));
    DevState init = *r->hst;
    const int iter0 = init.status == LPX_RUNNING || init.iter > 0 ? init.iter : 0;
    init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = 2; init.iter = iter0;
    init.dual_iter = iter0 + iters;                       // the iteration cap travels in the record
    if (init.pad[0] <= 0) init.pad[0] = r->n;
    *r->hst = init;
    LPX_HIP_TRY(hipMemcpy(r->st, r->hst, sizeof(DevState), hipMemcpyHostToDevice));
    std::vector<hipEvent_t> ev((size_t)8 * iters);
    for (hipEvent_t& e : ev) LPX_HIP_TRY(hipEventCreate(&e));
    const int chunks = (p.m + 127) / 128, per = (chunks + RVF_NW - 1) / RVF_NW;
    for (int i = 0; i < iters; ++i) {
        hipEvent_t* e = ev.data() + (size_t)8 * i;
        if (per <= 1) rv_launch_fused_profiled<1>(p, r->stream, e); else if (per <= 2) rv_launch_fused_profiled<2>(p, r->stream, e);
        else if (per <= 4) rv_launch_fused_profiled<4>(p, r->stream, e); else rv_launch_fused_profiled<8>(p, r->stream, e);
    }
    hipError_t le = hipGetLastError();
    hipError_t se = hipStreamSynchronize(r->stream);
    int rc = 0;
    if (le != hipSuccess || se != hipSuccess) { set_error(std::string("lpx_revised_profile: ") + hipGetErrorString(le != hipSuccess ? le : se)); rc = LPX_EDEVICE; }
    double sum[4] = {0, 0, 0, 0};
    int done = 0;
    if (!rc) {
        if (hipMemcpy(r->hst, r->st, sizeof(DevState), hipMemcpyDeviceToHost) != hipSuccess) rc = LPX_EDEVICE;
        done = rc ? 0 : std::min(iters, r->hst->iter - iter0);
        for (int i = 0; i < done && !rc; ++i)
            for (int k = 0; k < 4; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[(size_t)8 * i + 2 * k], ev[(size_t)8 * i + 2 * k + 1]) != hipSuccess) { rc = LPX_EDEVICE; set_error("lpx_revised_profile: event timing failed"); break; }
                sum[k] += ms;
            }
    }
    for (hipEvent_t e : ev) hipEventDestroy(e);
    if (rc) return rc;
    for (int k = 0; k < 4; ++k) us[k] = done > 0 ? 1e3 * sum[k] / done : 0.0;
    if (measured) *measured = done;
    if (r->hst->status == LPX_ITER_LIMIT) {                // the cap of this profile run, not an outcome: the handle stays usable
        r->hst->status = LPX_RUNNING;
        LPX_HIP_TRY(hipMemcpy(r->st, r->hst, sizeof(DevState), hipMemcpyHostToDevice));
    }
    return rv_flush_pending(r);
}

int lpx_invert(const double* M, int n, double* inv)
{
    if (!M || !inv || n < 1) { set_error("lpx_invert: bad argument"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    InvWork w;
    rc = inv_alloc(w, n);
    if (rc) return rc;
    double* Md = nullptr;
    LPX_HIP_TRY(hipMalloc((void**)&Md, sizeof(double) * (size_t)n * n));
    hipError_t e = hipMemcpyAsync(Md, M, sizeof(double) * (size_t)n * n, hipMemcpyHostToDevice, w.stream);
    if (e == hipSuccess) { hipLaunchKernelGGL(inv_build, dim3((w.ld + 255) / 256, n), dim3(256), 0, w.stream, (const double*)Md, n, n, w.A, w.ld); e = hipGetLastError(); }
    if (e != hipSuccess) { hipFree(Md); set_error(std::string("lpx_invert: ") + hipGetErrorString(e)); return LPX_EDEVICE; }
    rc = inv_run(w);
    hipFree(Md);
    if (rc) return rc;
    LPX_HIP_TRY(hipMemcpy2D(inv, sizeof(double) * n, w.A + n, sizeof(double) * w.ld, sizeof(double) * n, n, hipMemcpyDeviceToHost));
    return 0;
}

int lpx_revised_result(lpx_revised* r, int32_t* Bidx, int32_t* Nidx, double* xB, double* z)
{
    if (!r) return LPX_EINVAL;
    const int m = r->m, n = r->n;
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    { const int rc = rv_flush_pending(r); if (rc) return rc; }
    if (Bidx) LPX_HIP_TRY(hipMemcpy(Bidx, r->Bidx, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    if (xB) LPX_HIP_TRY(hipMemcpy2D(xB, sizeof(double), r->W + m, sizeof(double) * r->ldw, sizeof(double), m, hipMemcpyDeviceToHost));
    if (z) LPX_HIP_TRY(hipMemcpy(z, r->W + (size_t)m * r->ldw + m, sizeof(double), hipMemcpyDeviceToHost));
    if (Nidx) {
        std::vector<int32_t> key(n + m);
        LPX_HIP_TRY(hipMemcpy(key.data(), r->key, sizeof(int32_t) * (n + m), hipMemcpyDeviceToHost));
        std::vector<std::pair<int32_t, int32_t>> nb;
        for (int j = 0; j < n + m; ++j) if (key[j] >= 0) nb.emplace_back(key[j], j);
        std::sort(nb.begin(), nb.end());
        for (int j = 0; j < n && j < (int)nb.size(); ++j) Nidx[j] = nb[j].second;
    }
    return 0;
}

int lpx_revised_binv(lpx_revised* r, double* Binv)
{
    if (!r || !Binv) return LPX_EINVAL;
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    { const int rc = rv_flush_pending(r); if (rc) return rc; }
    LPX_HIP_TRY(hipMemcpy2D(Binv, sizeof(double) * r->m, r->W, sizeof(double) * r->ldw, sizeof(double) * r->m, r->m, hipMemcpyDeviceToHost));
    return 0;
}

int lpx_revised_iteration_view(lpx_revised* r, double* rc, double* d)
{
    if (!r) return LPX_EINVAL;
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    if (rc) LPX_HIP_TRY(hipMemcpy(rc, r->rc, sizeof(double) * (r->n + r->m), hipMemcpyDeviceToHost));
    if (d) LPX_HIP_TRY(hipMemcpy(d, r->fac, sizeof(double) * r->m, hipMemcpyDeviceToHost));
    return 0;
}

int lpx_revised_trace(lpx_revised* r, int32_t* trace, int cap, int* n)
{
    if (!r) return LPX_EINVAL;
    LPX_HIP_TRY(hipStreamSynchronize(r->stream));
    LPX_HIP_TRY(hipMemcpy(r->hst, r->st, sizeof(DevState), hipMemcpyDeviceToHost));
    int k = r->hst->iter; if (k > r->trace_cap) k = r->trace_cap;
    if (n) *n = k;
    if (trace && cap > 0) { int c = k < cap ? k : cap; if (c > 0) LPX_HIP_TRY(hipMemcpy(trace, r->trace, sizeof(int32_t) * 2 * c, hipMemcpyDeviceToHost)); }
    return 0;
}

int lpx_revised_solve(const double* A, int m, int n, const double* c, const double* b,
                      int32_t* Bidx, int32_t* Nidx, double* xB, double* z,
                      double eps, int max_iter, lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    lpx_revised* r = nullptr;
    double t0 = now_ms();
    int rc = lpx_revised_create(m, n, A, c, b, &r);
    if (rc) return rc;
    double h2d = now_ms() - t0;
    lpx_run_opts o; lpx_default_opts(&o, 1); o.eps = eps; o.max_iter = max_iter;
    lpx_stats local; std::memset(&local, 0, sizeof(local));
    int status = lpx_revised_run(r, &o, cb, user, &local);
    if (status >= 0) {
        t0 = now_ms();
        rc = lpx_revised_result(r, Bidx, Nidx, xB, z);
        local.d2h_ms = now_ms() - t0; local.h2d_ms = h2d;
        if (st) *st = local;
    }
    lpx_revised_destroy(r);
    return (status >= 0 && rc) ? rc : status;
}

}  // extern "C"
