// lpx_resident.hip -- the primal simplex loop with the WHOLE tableau resident on chip.
//
// When the tableau fits the chip's aggregate LDS (256 CUs x 160 KB = 40 MB; config 2's 1025 x 3073 tableau
// is 25 MB) the streaming path is the wrong shape: it moves 16*R*C bytes through HBM / Infinity Cache per
// pivot and pays two kernel boundaries, although nothing but the pivot row and one ratio per row ever has
// to leave a CU.  Here one persistent workgroup per CU keeps `rpw` consecutive constraint rows in its LDS for
// the whole solve, plus its own replica of the objective row (every workgroup updates it redundantly, so the
// entering column of ChooseEntering, Models/PrimalSimplex.cs:205-220, is known everywhere without any
// communication).  Per pivot exactly two things cross CUs, both through self-validating tagged granules in
// global memory (no grid barrier, no fence):
//
//   1. ChooseLeaving (Models/PrimalSimplex.cs:222-243): each workgroup publishes the ratio rhs/a of its rows
//      (+inf when a <= eps); every workgroup gathers all m ratios and runs the exact hysteresis scan itself.
//   2. Pivot (Models/PrimalSimplex.cs:245-257): the owner of row r divides it by the pivot (true division) and
//      publishes it; every workgroup gathers it and updates its rows (`t - f*p`, multiply and subtract rounded
//      separately) -- bit for bit the arithmetic of lpx_update.
//
// Granule: a double travels as two naturally aligned 8-byte words {32 data bits, 32-bit generation tag},
// each written by ONE write-through (sc1) store and read by sc1 loads that bypass the reader's L1; a reader
// accepts a value only when both tags equal the generation it waits for, so no ordering between the words or
// against any flag is needed.  Exchange buffers are double-buffered on the generation's parity: a workgroup can
// run at most one exchange ahead of the slowest one (it needs that one's ratio to get the next pivot row).
// Every wait is bounded (RS_SPIN_MAX polls); on expiry the workgroup raises the abort flag in the state
// record and leaves without writing its rows back.  A workgroup that was scheduled late may still have finished a
// short launch and written its rows, so the host keeps a copy of the tableau as it was when the launch started and
// puts it back whenever the flag is up (lpx_tableau.cpp, run_resident); the streaming kernels continue from there.
#include "lpx_resident.h"
#include <cstdlib>

#ifndef RS_SLEEP_A
#define RS_SLEEP_A 15        // consumer's timed first look at the pivot row: A + B * ceil(C / 1024), x 64 cycles
#define RS_SLEEP_B 5
#endif

namespace lpx {

struct ResParams {
    double* T; int ld; int R; int C;            // live shape
    int rpw;                                    // constraint rows per workgroup
    int mcap;                                   // rows of one parity half of xr
    int32_t* basis; int32_t* trace; int trace_cap;
    DevState* st;
    unsigned long long* xr;                     // [2][mcap][2]  ratio granules
    unsigned long long* xp;                     // [2][ld][2]    pivot-row granules
    unsigned* xgen;                             // generation counter, survives launches
    double eps, tol;
    int max_iter, chunk;
    int mute;                                   // diagnostic (LPX_RESIDENT_TEST_MUTE=1|2): the last workgroup plays dead
    int defer;                                  // the update of pivot k is applied in round k+1, in front of the pivot row's arrival
};

#ifdef LPX_STAMPS
#define RS_T0 unsigned long long rs_prev_ = __builtin_amdgcn_s_memtime();
#define RS_T(slot) do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1 ? 1 : 0)) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); P.xp[4 * (size_t)P.ld + (slot)] += n_ - rs_prev_; rs_prev_ = n_; } } while (0)
#else
#define RS_T0
#define RS_T(slot) do {} while (0)
#endif

__global__ __launch_bounds__(RS_NT, 1) void lpx_resident_primal(ResParams P)
{
    extern __shared__ __align__(16) double rs_lds[];
    __shared__ double s_v[RS_NT / 64];
    __shared__ int s_i[RS_NT / 64];             // [0..3] partial argmins, [4..7] partial band counts
    __shared__ int s_out;

    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int t = threadIdx.x, w = blockIdx.x;
    if ((P.mute == 1 || (P.mute == 2 && st->iter > 0)) && w == (int)gridDim.x - 1) return;   // 2: from the second launch on
    // A workgroup that starts after another one of this launch has given up cannot finish either: leave at once (the host
    // restores the tableau of the launch's start whatever was written back, lpx_tableau.cpp run_resident).
    if (rs_abort_raised(st)) return;
    if (P.mute == 3 && w == (int)gridDim.x - 1) rs_wait_for_abort(st);                        // 3: a LATE workgroup
    const int ld = P.ld, C = P.C, m = P.R - 1, rpw = P.rpw;
    const int row0 = w * rpw;
    const int nloc = max(0, min(rpw, m - row0));
    double* tile = rs_lds;                      // [rpw][ld]
    double* obj = tile + (size_t)rpw * ld;      // [ld]   replica of the objective row
    double* prow = obj + ld;                    // [ld]   pivot row of the current pivot
    double* ratios = prow + ld;                 // [m]    gathered ratios
    double* fac0 = ratios + ((m + 1) & ~1);     // [rpw+1] column factors of the local rows, then of obj ...
    double* fac1 = fac0 + ((rpw + 2) & ~1);     // ... ping-pong: the next pivot's are written while this one's are in use

    for (int i = 0; i < nloc; ++i) {
        const double* src = P.T + (size_t)(row0 + i) * ld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = *reinterpret_cast<const double2*>(src + j);
    }
    {
        const double* src = P.T + (size_t)m * ld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            *reinterpret_cast<double2*>(obj + j) = *reinterpret_cast<const double2*>(src + j);
            *reinterpret_cast<double2*>(prow + j) = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    int iter = st->iter;
    unsigned gen = *P.xgen;
    int status = LPX_RUNNING;
    bool hung = false;
    int r = -1, qlast = -1;
    // ChooseEntering on the initial objective row, :205-220
    int q = block_first_min_below<RS_NT>(obj, 1, C - 1, P.eps, s_v, s_i);
    double* fac = fac0;                         // factors of the pivot about to be made
    double* facn = fac1;
    // ratios of the first pivot of this launch (:229-233); later ones are published by the lookahead below
    if (q >= 0 && iter < P.max_iter && P.chunk > 0) {
        if (t < nloc) {
            const double a = tile[(size_t)t * ld + q];
            const double rhs = tile[(size_t)t * ld + C - 1];
            fac[t] = a;
            rs_publish(P.xr + 2 * ((size_t)((gen + 1u) & 1u) * P.mcap + row0 + t), a > P.eps ? rhs / a : __builtin_inf(), gen + 1u);
        }
        if (t == RS_NT - 1) fac[rpw] = obj[q];
    }
    __syncthreads();

    // DEFERRED UPDATE (P.defer, r03): the rank-1 update of pivot k stays pending through the end of round k and the decision of round
    // k+1 and is applied where a workgroup used to sleep and poll -- between the decision and the arrival of pivot k+1's row.  Its
    // factors are the ones `facn` holds after the swap at the end of round k (the lookahead of round k+1 rewrites facn only later),
    // its row is `prow` (rewritten by the gather only after the update), its owner's row `pend_skip` is already normalised in place.
    // The owner of pivot k+1's row forms that row as the pending update would (mul, then sub: the bits the update stores), divides,
    // publishes, leaves the result in the tile -- the pending update then skips that row as well -- and copies it to prow afterwards.
    // Nothing else reads the tile in between: the lookahead runs behind the update as before.
    const bool defer = P.defer != 0;
    bool pend = false; int pend_skip = -1;
    RS_T0
    const int t_outer = t;
    for (int k = 0; k < P.chunk; ++k) {
        int t = t_outer;                        // opaque per-round copy: keeps lane-dependent addresses from being
        asm volatile("" : "+v"(t));              // hoisted out of the round loop and held in registers (see lpx_resident_group.hip)
        if (iter >= P.max_iter) { status = LPX_ITER_LIMIT; break; }            // :95-96
        if (q < 0) { status = LPX_OPTIMAL; break; }                             // :99
        ++gen;
        const int par = (int)(gen & 1u);
        // ---- exchange 1: gather the ratios of all rows ------------------------------------------------
        int fail = 0;
        for (int base = t; base < m; base += RS_NT * RS_FETCH) {
            int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < m) cnt = u + 1; }
            if (!rs_gather(P.xr + 2 * (size_t)par * P.mcap, idx, cnt, gen, val)) fail = 1;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) ratios[idx[u]] = val[u];
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RS_T(1);
        // The hysteresis scan of :234-241, identical in every workgroup.  Fast path (lpx_block.h,
        // wave_hysteresis_argmin): with rmin the smallest ratio and i* its first row, if no OTHER row j has
        // fl(r_j - tol) <= rmin the sequential scan ends at i* whatever it accepted on the way.  Ties and
        // near-ties (the degenerate vertices of 0/1 programs) take the exact scan on wave 0.
        MinIdx lm; lm.v = __builtin_inf(); lm.i = INT_MAX;
        if (t < RS_RT)
            for (int i = t; i < m; i += RS_RT) { const double v = ratios[i]; if (v < lm.v) { lm.v = v; lm.i = i; } }
        lm = first4_min_idx(lm, s_v, s_i);
        if (lm.i == INT_MAX) { r = -1; }
        else {
            if (t < RS_RT) {
                int inband = 0;
                for (int i = t; i < m; i += RS_RT) inband += ((ratios[i] - P.tol) <= lm.v) ? 1 : 0;
                const int wsum = __popcll(__ballot(inband == 1)) + 2 * __popcll(__ballot(inband >= 2));
                if ((t & 63) == 0) s_i[4 + (t >> 6)] = wsum;
            }
            __syncthreads();
            if (s_i[4] + s_i[5] + s_i[6] + s_i[7] == 1) r = lm.i;
            else {
                if ((t >> 6) == 0) {
                    const int win = wave_hysteresis_argmin(m, P.tol, LdsRatio{ratios});
                    if (t == 0) s_out = win;
                }
                __syncthreads();
                r = s_out;
            }
        }
        if (r < 0) { status = LPX_UNBOUNDED; break; }                          // :102-106
        RS_T(2);
        // ---- exchange 2: the normalised pivot row, :247-249 ----------------------------------------
        const int owner = r / rpw, rl = r - owner * rpw;
        u64* xp = P.xp + 2 * (size_t)par * ld;
        if (w == owner) {
            double* prw = tile + (size_t)rl * ld;
            const bool fix = pend && rl != pend_skip;                           // row rl still lacks the pending update
            const double fp = fix ? facn[rl] : 0.0;
            double piv = prw[q];
            if (fix) { const double prod = fp * prow[q]; piv = piv - prod; }
            __syncthreads();
            for (int j = t; j < C; j += RS_NT) {
                double x = prw[j];
                if (fix) { const double prod = fp * prow[j]; x = x - prod; }
                const double p = x / piv;
                rs_publish(xp + 2 * (size_t)j, p, gen);
                prw[j] = p;
                if (!defer) prow[j] = p;                                        // defer: prow is still the pending update's row
            }
            if (defer && fix)                                                   // the padding columns of the row get the pending update too
                for (int j = C + t; j < ld; j += RS_NT) { const double prod = fp * prow[j]; prw[j] = prw[j] - prod; }
        }
        if (defer) {
            const bool had = pend;
            if (pend) {
                const int skip2 = (w == owner) ? rl : -1;
                for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
                    const double2 p = *reinterpret_cast<const double2*>(prow + j);
                    for (int i = 0; i < nloc; ++i) {
                        if (i == pend_skip || i == skip2) continue;
                        const double f = facn[i];
                        double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                        double prod = f * p.x; v.x = v.x - prod;
                        prod = f * p.y; v.y = v.y - prod;
                        *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
                    }
                }
                pend = false;
            }
            rs_barrier_lds();                   // prow may be rewritten now
            if (w == owner) { const double* prw = tile + (size_t)rl * ld; for (int j = t; j < C; j += RS_NT) prow[j] = prw[j]; }
            else if (!had) { __builtin_amdgcn_s_sleep(RS_SLEEP_A); for (int z = 0; z < C; z += RS_NT) __builtin_amdgcn_s_sleep(RS_SLEEP_B); }
        }
        if (w != owner) {
            // Poll ONE granule (the last column, covered by the owner's last store instruction) until the row is
            // on its way: 255 workgroups re-reading 48 KB each per failed poll would saturate the fabric.
            // The owner needs 0.5-1 us to divide and store the row: sleep through that, then try the
            // whole gather ONCE -- when the row is already visible this saves the canary's round trip.
            if (!defer) {
                __builtin_amdgcn_s_sleep(RS_SLEEP_A);
                for (int z = 0; z < C; z += RS_NT) __builtin_amdgcn_s_sleep(RS_SLEEP_B);   // measured best: 20 (C = 769) ... 35 (C = 3073) x 64 cycles
            }
            bool first = true;
            for (int base = t; base < C; base += RS_NT * RS_FETCH) {
                int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < C) cnt = u + 1; }
                unsigned pend = (1u << cnt) - 1u;
                if (first && !rs_gather(xp, idx, cnt, gen, val, 1u, &pend)) {
                    if (!rs_wait(xp + 2 * (size_t)(C - 1), gen)) fail = 1;
                }
                first = false;
                if (pend && !rs_gather(xp, idx, cnt, gen, val, RS_SPIN_MAX, &pend)) fail = 1;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) prow[idx[u]] = val[u];
            }
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RS_T(3);
        // ---- objective replica first (:250-256 for row m): it names the NEXT entering column ---------------
        const double fobj = fac[rpw];
        const int skip = (w == owner) ? rl : -1;
        MinIdx best; best.v = -P.eps; best.i = INT_MAX;
        if (t < RS_RT) {
            for (int j = 2 * t; j < ld; j += 2 * RS_RT) {
                const double2 p = *reinterpret_cast<const double2*>(prow + j);
                double2 o = *reinterpret_cast<double2*>(obj + j);
                double prod = fobj * p.x; o.x = o.x - prod;
                prod = fobj * p.y; o.y = o.y - prod;
                *reinterpret_cast<double2*>(obj + j) = o;
                if (j < C - 1 && o.x < best.v) { best.v = o.x; best.i = j; }
                if (j + 1 < C - 1 && o.y < best.v) { best.v = o.y; best.i = j + 1; }
            }
        }
        if (w == 0 && t == 0) {
            P.basis[r] = q;                                                     // :110
            if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        }
        qlast = q;
        ++iter;
        best = first4_min_idx(best, s_v, s_i);                                  // next ChooseEntering, :205-220
        const int qn = best.i == INT_MAX ? -1 : best.i;
        RS_T(4);
        // ---- lookahead: the ratios of the next pivot leave BEFORE the bulk of the update, so that their trip
        //      overlaps it.  a' and rhs' are the values the update below is about to store (same mul, same sub).
        if (qn >= 0 && iter < P.max_iter && k + 1 < P.chunk) {
            if (t < nloc) {
                double a, rhs;
                if (t == skip) { a = prow[qn]; rhs = prow[C - 1]; }
                else {
                    const double f = fac[t];
                    double prod = f * prow[qn]; a = tile[(size_t)t * ld + qn] - prod;
                    prod = f * prow[C - 1]; rhs = tile[(size_t)t * ld + C - 1] - prod;
                }
                facn[t] = a;
                rs_publish(P.xr + 2 * ((size_t)(par ^ 1) * P.mcap + row0 + t), a > P.eps ? rhs / a : __builtin_inf(), gen + 1u);
            }
            if (t == RS_NT - 1) facn[rpw] = obj[qn];
        }
        // the lookahead read columns qn and C-1 before anyone rewrites them.  LDS only: a full barrier would also wait for the
        // acknowledgements of the lookahead's write-through stores, which nobody in this workgroup needs (lpx_resident.h)
        rs_barrier_lds();
        RS_T(0);
        // ---- rank-1 update of the local rows, :250-256 -------------------------------------------------------
        if (defer) { pend = true; pend_skip = skip; }
        else
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            const double2 p = *reinterpret_cast<const double2*>(prow + j);
            for (int i = 0; i < nloc; ++i) {
                if (i == skip) continue;
                const double f = fac[i];
                double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                double prod = f * p.x; v.x = v.x - prod;
                prod = f * p.y; v.y = v.y - prod;
                *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
            }
        }
        { double* sw = fac; fac = facn; facn = sw; }
        q = qn;
        rs_barrier_lds();
        RS_T(5);
    }

    if (hung) {
        if (t == 0) atomicOr(&st->pad[1], 1);
        return;
    }
    if (pend) {                                 // the last pivot's update, on the way out (its factors: facn, see the swap)
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            const double2 p = *reinterpret_cast<const double2*>(prow + j);
            for (int i = 0; i < nloc; ++i) {
                if (i == pend_skip) continue;
                const double f = facn[i];
                double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                double prod = f * p.x; v.x = v.x - prod;
                prod = f * p.y; v.y = v.y - prod;
                *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
            }
        }
        __syncthreads();
    }
    for (int i = 0; i < nloc; ++i) {
        double* dst = P.T + (size_t)(row0 + i) * ld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(tile + (size_t)i * ld + j);
    }
    if (w == 0) {
        double* dst = P.T + (size_t)m * ld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(obj + j);
        if (t == 0) {
            st->status = status; st->iter = iter; st->primal_count = iter;
            st->r = r; st->q = qlast;
            *P.xgen = gen;
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------
static size_t resident_lds_bytes(int R, int C, int ld, int rpw)
{
    const int m = R - 1;
    return sizeof(double) * ((size_t)(rpw + 2) * ld + ((m + 1) & ~1) + 2 * (size_t)((rpw + 2) & ~1) + 2);
}

// Picks the grid for a live shape; returns 0 when the tableau does not fit on chip.
int resident_plan(int R, int C, int ld, int* grid, int* rpw, size_t* lds)
{
    static int cus = 0;
    static size_t lds_max = 0;
    if (!cus) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        cus = prop.multiProcessorCount;
        lds_max = 160 * 1024 - 1024;            // one workgroup may take the CU's whole LDS (gfx950: 160 KB)
    }
    const int m = R - 1;
    if (m < 1 || cus < 1) return 0;
    const int g = m < cus ? m : cus;
    const int rp = (m + g - 1) / g;
    const int gg = (m + rp - 1) / rp;
    const size_t need = resident_lds_bytes(R, C, ld, rp);
    if (need > lds_max) return 0;
    *grid = gg; *rpw = rp; *lds = need;
    return 1;
}

hipError_t resident_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lpx_resident_primal),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}

hipError_t launch_resident_primal(double* T, int ld, int R, int C, int grid, int rpw, size_t lds, int mcap,
                                  int32_t* basis, int32_t* trace, int trace_cap, DevState* st,
                                  unsigned long long* xr, unsigned long long* xp, unsigned* xgen,
                                  double eps, double tol, int max_iter, int chunk, hipStream_t s)
{
    ResParams p;
    p.T = T; p.ld = ld; p.R = R; p.C = C; p.rpw = rpw; p.mcap = mcap;
    p.basis = basis; p.trace = trace; p.trace_cap = trace_cap; p.st = st;
    p.xr = xr; p.xp = xp; p.xgen = xgen; p.eps = eps; p.tol = tol; p.max_iter = max_iter; p.chunk = chunk;
    static const int mute = [] { const char* e = std::getenv("LPX_RESIDENT_TEST_MUTE"); return e ? std::atoi(e) : 0; }();
    p.mute = mute;
    static const int defer = [] { const char* e = std::getenv("LPX_RESIDENT_DEFER"); return (e && e[0] == '0') ? 0 : 1; }();   // diagnostic: 0 = update at the end of its own round
    p.defer = defer;
    hipLaunchKernelGGL(lpx_resident_primal, dim3(grid), dim3(RS_NT), lds, s, p);
    return hipGetLastError();
}

}  // namespace lpx
