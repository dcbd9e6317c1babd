// lpx_resident_col.hip -- the resident primal loop with COLUMN-owning workgroups: one exchange per pivot.
//
// lpx_resident.hip keeps rows on each CU and pays two dependent cross-CU exchanges per pivot: all m ratios (every
// workgroup needs them for ChooseLeaving) and then the normalised pivot row (3073 values at config 2).  Here workgroup w owns
// `cpw` consecutive COLUMNS of all R rows (column-major in LDS) plus a replica of the RHS column, which every workgroup
// updates redundantly.  Then
//   * the pivot row is local: p_j = T[r,j] / piv for the owned columns (Models/PrimalSimplex.cs:247-250), no exchange;
//   * ChooseLeaving (:222-243) needs only the entering column q: ratio_i = rhs_i / T[i,q] from the RHS replica, then the exact
//     hysteresis scan of lpx_resident.h, identical in every workgroup;
//   * ChooseEntering (:205-220) is a first-index argmin over the objective entries, which are spread over the workgroups:
//     each one publishes its best candidate (value, column) AND, speculatively, that column itself as the update is about to
//     leave it (the lookahead of lpx_resident.hip: same mul, same sub as the bulk update).  Whoever wins, its column is already
//     on its way when the candidates are compared -- candidates and pivot column cost ONE exchange latency, not two.
// Per pivot and workgroup: 2 + R tagged granules stored, 2 G + R gathered (G = workgroups), against R/G + C stored and m + C
// gathered by the row-owning kernel.  Arithmetic per element is that of lpx_update, so results are bit-identical.
// Granules, generations, double buffering on the generation's parity, bounded waits, the abort flag and the host-side
// restore are those of lpx_resident.hip (lpx_resident.h); a workgroup can run at most one exchange ahead of the slowest,
// because it needs that one's candidate to start the next pivot.
#include "lpx_resident.h"
#include <cstdlib>

namespace lpx {

struct ResColParams {
    double* T; int ld; int R; int C;            // live shape
    int cpw;                                    // columns per workgroup
    int Rp;                                     // R rounded up to even: column stride in LDS and in xq
    int32_t* basis; int32_t* trace; int trace_cap;
    DevState* st;
    unsigned long long* xc;                     // [2][G][2][2]    candidate granule pairs {value, column}
    unsigned long long* xq;                     // [2][G][Rp][2]   the candidate's column, granule pairs
    unsigned* xgen;                             // generation counter, survives launches
    double eps, tol;
    int max_iter, chunk;
    int mute;                                   // diagnostic, LPX_RESIDENT_TEST_MUTE (see lpx_resident.hip)
};

#ifdef LPX_STAMPS
// diagnostic build only: workgroup 1 lane 0 accumulates s_memtime deltas per phase behind the candidate granules
#define RC_T0 unsigned long long rc_prev_ = __builtin_amdgcn_s_memtime();
#define RC_T(slot) do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1 ? 1 : 0)) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); P.xc[2 * 256 * 2 * 2 + (slot)] += n_ - rc_prev_; rc_prev_ = n_; } } while (0)
#else
#define RC_T0
#define RC_T(slot) do {} while (0)
#endif

__global__ __launch_bounds__(RS_NT, 1) void lpx_resident_primal_col(ResColParams P)
{
    extern __shared__ __align__(16) double rs_lds[];
    __shared__ double s_v[RS_NT / 64];
    __shared__ int s_i[RS_NT / 64];
    __shared__ int s_out;

    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    if ((P.mute == 1 || (P.mute == 2 && st->iter > 0)) && w == G - 1) return;
    if (rs_abort_raised(st)) return;
    if (P.mute == 3 && w == G - 1) rs_wait_for_abort(st);
    const int R = P.R, C = P.C, m = R - 1, Rp = P.Rp, cpw = P.cpw, ld = P.ld;
    const int c0 = w * cpw;
    const int ncol = max(0, min(cpw, C - c0));
    double* tile = rs_lds;                      // [cpw][Rp]  owned columns, column-major
    double* rhs = tile + (size_t)cpw * Rp;      // [Rp]       replica of the RHS column
    double* colq = rhs + Rp;                    // [Rp]       entering column of the current pivot (factors)
    double* ratios = colq + Rp;                 // [Rp]
    double* pbuf = ratios + Rp;                 // [cpw]      normalised pivot-row entries of the owned columns
    double* cand = pbuf + ((cpw + 1) & ~1);     // [2 G]      gathered candidates

    for (int i = t; i < R; i += RS_NT) {
        const double* src = P.T + (size_t)i * ld;
        for (int c = 0; c < ncol; ++c) tile[(size_t)c * Rp + i] = src[c0 + c];
        rhs[i] = src[C - 1];
    }
    __syncthreads();

    int iter = st->iter;
    unsigned gen = *P.xgen;
    int status = LPX_RUNNING;
    bool hung = false;
    int r = -1, qlast = -1;

    // publishes this workgroup's candidate for generation g: best objective entry among the owned columns (value < -eps,
    // first index on ties; the RHS column never enters, :208) and that column as it stands in the tile
    auto publish = [&](MinIdx best, unsigned g) {
        const int par = (int)(g & 1u);
        const int cs = best.i == INT_MAX ? -1 : best.i - c0;
        if (cs >= 0) {
            unsigned long long* dst = P.xq + 2 * ((size_t)(par * G + w) * Rp);
            for (int i = t; i < R; i += RS_NT) rs_publish(dst + 2 * (size_t)i, tile[(size_t)cs * Rp + i], g);
        }
        if (t == 0) {
            unsigned long long* dc = P.xc + 2 * ((size_t)(par * G + w) * 2);
            rs_publish(dc, cs >= 0 ? best.v : __builtin_inf(), g);
            rs_publish(dc + 2, cs >= 0 ? (double)best.i : 2147483647.0, g);
        }
    };
    auto local_candidate = [&]() {
        MinIdx b; b.v = -P.eps; b.i = INT_MAX;
        if (t < RS_RT)
            for (int c = t; c < ncol; c += RS_RT) {
                const int col = c0 + c;
                if (col < C - 1) { const double o = tile[(size_t)c * Rp + m]; if (o < b.v) { b.v = o; b.i = col; } }
            }
        return first4_min_idx(b, s_v, s_i);
    };
    if (iter < P.max_iter && P.chunk > 0) {
        const MinIdx b0 = local_candidate();
        __syncthreads();
        publish(b0, gen + 1u);
    }

    RC_T0
    const int t_outer = t;
    for (int k = 0; k < P.chunk; ++k) {
        int t = t_outer;                        // opaque per-round copy (see lpx_resident_group.hip)
        asm volatile("" : "+v"(t));
        if (iter >= P.max_iter) { status = LPX_ITER_LIMIT; break; }            // :95-96
        ++gen;
        const int par = (int)(gen & 1u);
        // ---- the one exchange: every workgroup's candidate ... -------------------------------------------------
        int fail = 0;
        for (int base = t; base < 2 * G; base += RS_NT * RS_FETCH) {
            int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < 2 * G) cnt = u + 1; }
            if (!rs_gather(P.xc + 2 * ((size_t)par * G * 2), idx, cnt, gen, val)) fail = 1;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) cand[idx[u]] = val[u];
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RC_T(0);
        // ChooseEntering, :205-220: first index of the strict minimum below -eps == lexicographic min of (value, column)
        MinIdx be; be.v = __builtin_inf(); be.i = INT_MAX;
        if (t < RS_RT)
            for (int g = t; g < G; g += RS_RT) {
                MinIdx x; x.v = cand[2 * g]; x.i = (int)cand[2 * g + 1];
                be = mi_pick(be, x);
            }
        be = first4_min_idx(be, s_v, s_i);
        const int q = (be.i == INT_MAX || !(be.v < -P.eps)) ? -1 : be.i;
        if (q < 0) { status = LPX_OPTIMAL; break; }                             // :99
        RC_T(1);
        // ---- ... and the winner's column, published together with its candidate ---------------------------------
        {
            const int owner = q / cpw;
            const unsigned long long* src = P.xq + 2 * ((size_t)(par * G + owner) * Rp);
            for (int base = t; base < R; base += RS_NT * RS_FETCH) {
                int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < R) cnt = u + 1; }
                if (!rs_gather(src, idx, cnt, gen, val)) fail = 1;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) colq[idx[u]] = val[u];
            }
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RC_T(2);
        // ---- ChooseLeaving, :222-243: ratios from the RHS replica, exact hysteresis scan (identical everywhere) ----------
        for (int i = t; i < m; i += RS_NT) { const double a = colq[i]; ratios[i] = a > P.eps ? rhs[i] / a : __builtin_inf(); }
        __syncthreads();
        r = rs_hysteresis(m, P.tol, ratios, s_v, s_i, &s_out);
        if (r < 0) { status = LPX_UNBOUNDED; break; }                          // :102-106
        RC_T(3);
        // ---- Pivot, :245-257, on the owned columns: the pivot row is local ------------------------------------------------
        const double piv = colq[r];
        const double fm = colq[m];
        for (int c = t; c < ncol; c += RS_NT) pbuf[c] = tile[(size_t)c * Rp + r] / piv;    // true division, :250
        const double prhs = rhs[r] / piv;
        __syncthreads();
        // next entering candidate: the objective entries as the update is about to leave them (same mul, same sub)
        MinIdx best; best.v = -P.eps; best.i = INT_MAX;
        if (t < RS_RT)
            for (int c = t; c < ncol; c += RS_RT) {
                const int col = c0 + c;
                if (col < C - 1) {
                    const double prod = fm * pbuf[c];
                    const double o = tile[(size_t)c * Rp + m] - prod;
                    if (o < best.v) { best.v = o; best.i = col; }
                }
            }
        best = first4_min_idx(best, s_v, s_i);
        const int cs = best.i == INT_MAX ? -1 : best.i - c0;
        RC_T(4);
        if (w == 0 && t == 0) {
            P.basis[r] = q;                                                     // :110
            if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        }
        qlast = q;
        ++iter;
        const bool more = iter < P.max_iter && k + 1 < P.chunk;
        // the candidate's column first: updated in place (exactly what the bulk update below would store) and sent off
        if (cs >= 0) {
            const double pc = pbuf[cs];
            unsigned long long* dst = P.xq + 2 * ((size_t)((par ^ 1) * G + w) * Rp);
            const int cso = cs * Rp;
            for (int i = t; i < R; i += RS_NT) {
                const double prod = colq[i] * pc;
                const double v = tile[cso + i] - prod;
                const double u = (i == r) ? pc : v;
                tile[cso + i] = u;
                if (more) rs_publish(dst + 2 * (size_t)i, u, gen + 1u);
            }
        }
        if (more && t == 0) {
            unsigned long long* dc = P.xc + 2 * ((size_t)((par ^ 1) * G + w) * 2);
            rs_publish(dc, cs >= 0 ? best.v : __builtin_inf(), gen + 1u);
            rs_publish(dc + 2, cs >= 0 ? (double)best.i : 2147483647.0, gen + 1u);
        }
        RC_T(5);
        // ---- bulk: the other owned columns and the RHS replica; its time hides behind the others' candidates -----------------
        {
            // LDS bandwidth is the limit here (128 B / clk / CU): per element one 8-byte read and one 8-byte write of the tile;
            // the row's factor stays in a register and the pivot-row entries are read eight at a time (one address per wave).
            // Constraint rows on lane = row; the objective row (R = m + 1 would give lane 0 a second pass) on the last wave,
            // lane = column.
            for (int i = t; i < m; i += RS_NT) {
                const bool isr = (i == r);
                const double f = colq[i];
                for (int cb = 0; cb < ncol; cb += 8) {
                    double pc[8], v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = min(cb + u, ncol - 1);
                        pc[u] = pbuf[c];
                        v[u] = tile[c * Rp + i];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = cb + u;
                        if (c < ncol && c != cs) {
                            const double prod = f * pc[u];
                            const double x = v[u] - prod;
                            tile[c * Rp + i] = isr ? pc[u] : x;       // row r takes the normalised value itself (:249-250)
                        }
                    }
                }
                const double prod = f * prhs;
                const double x = rhs[i] - prod;
                rhs[i] = isr ? prhs : x;
            }
            if (t >= RS_NT - 64) {
                const int l = t - (RS_NT - 64);
                for (int c = l; c < ncol; c += 64)
                    if (c != cs) { const double prod = fm * pbuf[c]; tile[c * Rp + m] = tile[c * Rp + m] - prod; }
                if (l == 0) { const double prod = fm * prhs; rhs[m] = rhs[m] - prod; }
            }
        }
        RC_T(7);
        rs_barrier_lds();                       // LDS only: the published granules' store acknowledgements are not waited for
        RC_T(6);
    }

    if (hung) {
        if (t == 0) atomicOr(&st->pad[1], 1);
        return;
    }
    for (int i = t; i < R; i += RS_NT) {
        double* dst = P.T + (size_t)i * ld;
        for (int c = 0; c < ncol; ++c) dst[c0 + c] = tile[(size_t)c * Rp + i];
    }
    if (w == 0 && t == 0) {
        st->status = status; st->iter = iter; st->primal_count = iter;
        st->r = r; st->q = qlast;
        *P.xgen = gen;
    }
}

// ---- host side --------------------------------------------------------------------------------------------
static size_t resident_col_lds(int R, int cpw, int grid)
{
    const size_t Rp = (size_t)((R + 1) & ~1);
    return sizeof(double) * ((size_t)(cpw + 3) * Rp + (size_t)((cpw + 1) & ~1) + 2 * (size_t)grid + 16);
}

// Picks the grid of the column-owning kernel for a live shape; 0 when a workgroup's columns do not fit its LDS.
int resident_col_plan(int R, int C, int* grid, int* cpw, size_t* lds)
{
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t prop; int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
    }
    if (R < 2 || C < 2 || cus < 1) return 0;
    const int g = C < cus ? C : cus;
    const int cp = (C + g - 1) / g;
    const int gg = (C + cp - 1) / cp;
    const size_t need = resident_col_lds(R, cp, gg);
    if (need > (size_t)(160 * 1024 - 1024)) return 0;
    *grid = gg; *cpw = cp; *lds = need;
    return 1;
}

size_t resident_col_xc_bytes(int grid) { return sizeof(unsigned long long) * (2 * (size_t)grid * 2 * 2 + 64); }   // + diagnostic stamps
size_t resident_col_xq_bytes(int grid, int R) { return sizeof(unsigned long long) * 2 * (size_t)grid * (size_t)((R + 1) & ~1) * 2; }

hipError_t resident_col_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lpx_resident_primal_col),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}

hipError_t launch_resident_primal_col(double* T, int ld, int R, int C, int grid, int cpw, size_t lds,
                                      int32_t* basis, int32_t* trace, int trace_cap, DevState* st,
                                      unsigned long long* xc, unsigned long long* xq, unsigned* xgen,
                                      double eps, double tol, int max_iter, int chunk, hipStream_t s)
{
    ResColParams p;
    p.T = T; p.ld = ld; p.R = R; p.C = C; p.cpw = cpw; p.Rp = (R + 1) & ~1;
    p.basis = basis; p.trace = trace; p.trace_cap = trace_cap; p.st = st;
    p.xc = xc; p.xq = xq; p.xgen = xgen; p.eps = eps; p.tol = tol; p.max_iter = max_iter; p.chunk = chunk;
    static const int mute = [] { const char* e = std::getenv("LPX_RESIDENT_TEST_MUTE"); return e ? std::atoi(e) : 0; }();
    p.mute = mute;
    hipLaunchKernelGGL(lpx_resident_primal_col, dim3(grid), dim3(RS_NT), lds, s, p);
    return hipGetLastError();
}

}  // namespace lpx
