// lpx_resident_regs.hip -- the resident group loop of lpx_resident_group.hip with the node's rows in REGISTERS instead of LDS.
//
// Cold branch-and-bound nodes (every node re-solved from the slack basis as the reference does, Models/Branch&Bound.cs:148) are
// bound by what fits on chip: four 7.9 MB tableaux fill the 40 MB of LDS, and the loop is latency bound (two cross-CU exchanges per
// pivot), so the chip idles through most of a step.  The register files hold three times as much as the LDS (512 KB per CU,
// 128 MB in all).  Here workgroup w of a node keeps RPW consecutive rows in VGPRs: lane t owns KC adjacent columns (KC t .. KC t + KC - 1)
// of every one of them (a workgroup is as wide as the tableau: NT >= ld / KC lanes), so
//   * the rank-1 update is register arithmetic (row factor broadcast from LDS, the lane's pivot-row pair read once per pivot);
//   * a COLUMN of the local rows (entering column -> factors; RHS and next entering column -> the ratios the rows publish) is
//     the RPW registers of ONE lane, which drops them into LDS;
//   * the pivot ROW of the owner is register `rl` of every lane, `rl` workgroup-uniform but dynamic: an unrolled select chain.
// With fewer lanes per CU a lane gets more registers, and fewer lanes pay the loop's own ~85 registers:
//   NT = 768, two columns per lane, 3 waves per SIMD (168 VGPRs): 22 rows of a 1282-column node per workgroup -> 35 workgroups per
//     node -> SEVEN config-4 nodes in flight instead of the LDS form's four (845-880 nodes/s; 26 rows / 8 nodes 802, 28 rows / 9 nodes
//     729: beyond 22 rows that tile spills into scratch on the critical path);
//   NT = 512, THREE columns per lane, 2 waves per SIMD (256 VGPRs): 28 rows -> 28 workgroups per node -> NINE nodes in flight: 964
//     nodes/s on the box where the first form gives 845 (31 rows / ten nodes spill 40 registers: 805);
//   the same with the 120 KB of LDS the row buffers leave free holding 11-12 MORE rows of the node (swept K0b-style after the register
//     rows: +1.5 us per step): 26 + 11 rows -> 21 workgroups per node -> TWELVE nodes in flight: 1170-1180 nodes/s (24-26 register rows
//     alike; 27: 1046, 28: 1024 -- the loop's own registers spill).
//   and with the update of pivot k DEFERRED into round k+1's wait for the pivot row (template parameter DEFER, the round loop's
//     comment): +8 % again.  A group of more nodes than the chip holds is ONE launch (lpx_tableau.cpp, run_resident_group): the hardware
//     starts the next node's workgroups as an earlier node's leave (+11 %: 1.31 k nodes/s on the cold config-4 search).
// Everything else -- the tagged-granule
// exchanges, the replicated state machine of the dual path, the lookahead, the bounded waits, the arithmetic per element -- is that of
// lpx_resident_group (bit-identical results; the same tests).  LDS holds the objective replica, the pivot row, the gathered column,
// the owner's row (for the dual loop's column scan) and the small column buffers: ~40 KB; the rest the node's LDS rows.
#include "lpx_resident.h"
#include <algorithm>
#include <cstdlib>

namespace lpx {

struct ResGroupParamsR { const ResNode* nodes; int chunk; int mute; int rt; };   // rt = rows a workgroup may hold (registers + LDS): sizes the per-row LDS arrays

#ifdef LPX_STAMPS
#define RR_T0 unsigned long long rg_prev_ = __builtin_amdgcn_s_memtime();
#define RR_T(slot) do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1 ? 1 : 0) && blockIdx.y == 0) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); P.xp[4 * ((size_t)P.ld + 8) + (slot)] += n_ - rg_prev_; rg_prev_ = n_; } } while (0)
#else
#define RR_T0
#define RR_T(slot) do {} while (0)
#endif

// rs_hysteresis (lpx_resident.h) with the exact replay INLINE and four ratios per lane and segment: the replay of lpx_resident.h is a
// real call, and a call makes the caller save its live registers -- here the whole register tile (28 dwordx4 stores and loads through
// scratch per call, on the critical path of almost every pivot of a 0/1 program: its ratios tie all the time).
__device__ __forceinline__ int rr_hysteresis(int m, double tol, const double* ratios, double* s_v, int* s_i, int* s_out)
{
    const int t = threadIdx.x;
    MinIdx lm; lm.v = __builtin_inf(); lm.i = INT_MAX;
    if (t < RS_RT)
        for (int i = t; i < m; i += RS_RT) { const double v = ratios[i]; if (v < lm.v) { lm.v = v; lm.i = i; } }
    lm = first4_min_idx(lm, s_v, s_i);
    int r;
    if (lm.i == INT_MAX) { r = -1; __syncthreads(); }
    else {
        if (t < RS_RT) {
            int inband = 0;
            for (int i = t; i < m; i += RS_RT) inband += ((ratios[i] - tol) <= lm.v) ? 1 : 0;
            const int wsum = __popcll(__ballot(inband == 1)) + 2 * __popcll(__ballot(inband >= 2));
            if ((t & 63) == 0) s_i[4 + (t >> 6)] = wsum;
        }
        __syncthreads();
        if (s_i[4] + s_i[5] + s_i[6] + s_i[7] == 1) r = lm.i;
        else {
            if ((t >> 6) == 0) {
                const int win = wave_hysteresis_argmin<LdsRatio, true, 4>(m, tol, LdsRatio{ratios});
                if (t == 0) *s_out = win;
            }
            __syncthreads();
            r = *s_out;
        }
        __syncthreads();
    }
    return r;
}

template <int NT>
__device__ __forceinline__ int rr_first_min_below(const double* v, int L, double eps, double* s_v, int* s_i)
{
    MinIdx m; m.v = -eps; m.i = INT_MAX;
    if (threadIdx.x < RS_RT)
        for (int j = threadIdx.x; j < L; j += RS_RT) { const double x = v[j]; if (x < m.v) { m.v = x; m.i = j; } }
    m = first4_min_idx(m, s_v, s_i);
    __syncthreads();
    return m.i == INT_MAX ? -1 : m.i;
}

// width of the register tile and of the LDS rows: the first multiple of 2 and of kc at or above C
__host__ __device__ __forceinline__ int rr_tile_width(int C, int kc) { const int u = (kc & 1) ? 2 * kc : kc; return (C + u - 1) / u * u; }

// KG granule pairs per lane and round (i, i + NT, [i + 2 NT]): two for the two-column form, three for the three-column form, so that
// a workgroup covers the pivot row (as wide as its tile) in ONE round trip
template <int KG, int NT>
__device__ __forceinline__ bool rr_gather(const u64* g, int base, int cnt, unsigned gen, double* val, unsigned max_spin = RS_SPIN_MAX, unsigned* pend = nullptr)
{
    if (KG == 3) return rs_gather3(g, base, base + NT, base + 2 * NT, cnt, gen, val, max_spin, pend);
    return rs_gather2(g, base, base + NT, cnt, gen, val, max_spin, pend);
}

// WPE = waves per SIMD the kernel is compiled for (NT / 256): sets the register budget.  KC = columns per lane: 2 for tableaux up to
// 2 NT columns; 3 lets 512 lanes (two waves per SIMD, 256 VGPRs each) carry a 1536-column node -- fewer lanes pay the loop's own ~80
// registers, so a CU holds 28 rows of a config-4 node instead of 22 and the chip NINE nodes instead of seven.
// ---- rank-1 update of the local rows (registers, then the rows in LDS) with the factors in `fac` and the normalised row in `prow`, :250-256;
//      `skip_` = the local row that IS the pivot row (owner only, else -1): it is put back afterwards from the normalised row (a predicate per
//      row would cost two scalar mask registers each); rows beyond nloc hold dummies nobody reads.  A macro, not a lambda: a lambda that
//      WRITES the register tile through its capture moved the whole tile into scratch memory.
#define RR_UPDATE(skip_) do { const int rr_sk = (skip_); \
    if (mine) { \
        double p[KC]; \
_Pragma("unroll") \
        for (int c = 0; c < KC; ++c) p[c] = prow[jt + c]; \
_Pragma("unroll") \
        for (int i = 0; i < RPW; ++i) { \
            const double f = fac[i]; \
_Pragma("unroll") \
            for (int c = 0; c < KC; ++c) { const double prod = f * p[c]; reg[i][c] = reg[i][c] - prod; } \
        } \
        if (rr_sk >= 0) { \
_Pragma("unroll") \
            for (int c = 0; c < KC; ++c) if (jt + c >= C) p[c] = rowbuf[jt + c]; \
_Pragma("unroll") \
            for (int i = 0; i < RPW; ++i) \
                if (i == rr_sk) { \
_Pragma("unroll") \
                    for (int c = 0; c < KC; ++c) reg[i][c] = p[c]; \
                } \
        } \
    } \
    if (nl > 0) { \
        for (int j = 2 * t; j < ld; j += 2 * NT) { \
            const double2 p2 = *reinterpret_cast<const double2*>(prow + j); \
            double* lr = lrows + j; \
            int i = 0; \
            for (; i + 2 <= nl; i += 2) { \
                double2 x0 = *reinterpret_cast<double2*>(lr + (size_t)i * ld), x1 = *reinterpret_cast<double2*>(lr + (size_t)(i + 1) * ld); \
                const double f0 = fac[RPW + i], f1 = fac[RPW + i + 1]; \
                double prod = f0 * p2.x; x0.x = x0.x - prod; prod = f0 * p2.y; x0.y = x0.y - prod; \
                prod = f1 * p2.x; x1.x = x1.x - prod; prod = f1 * p2.y; x1.y = x1.y - prod; \
                *reinterpret_cast<double2*>(lr + (size_t)i * ld) = x0; *reinterpret_cast<double2*>(lr + (size_t)(i + 1) * ld) = x1; \
            } \
            if (i < nl) { \
                double2 x0 = *reinterpret_cast<double2*>(lr + (size_t)i * ld); \
                const double f0 = fac[RPW + i]; \
                double prod = f0 * p2.x; x0.x = x0.x - prod; prod = f0 * p2.y; x0.y = x0.y - prod; \
                *reinterpret_cast<double2*>(lr + (size_t)i * ld) = x0; \
            } \
            if (rr_sk >= RPW) { \
                double2 v = p2; \
                if (j >= C) v.x = rowbuf[j]; \
                if (j + 1 >= C) v.y = rowbuf[j + 1]; \
                *reinterpret_cast<double2*>(lr + (size_t)(rr_sk - RPW) * ld) = v; \
            } \
        } \
    } \
} while (0)

// DEFER: the rank-1 update of pivot k is applied while the workgroup would otherwise wait for pivot k+1's row (see the round loop).
template <int NT, int RPW, int WPE, int KC, bool DEFER>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void lpx_resident_group_r(ResGroupParamsR GP)
{
    extern __shared__ __align__(16) double rr_lds[];
    __shared__ double s_v[16];
    __shared__ int s_i[16];
    __shared__ int s_out;

    const ResNode& N = GP.nodes[blockIdx.y];
    struct { double* T; int ld, R, C; unsigned long long* xr; unsigned long long* xp; int mcap, dual; } P;
    P.T = N.T; P.ld = N.ld; P.R = N.R; P.C = N.C; P.xr = N.xr; P.xp = N.xp; P.mcap = N.mcap; P.dual = N.dual;
    const double eps = N.eps;
    DevState* st = N.st;
    if (st->status != LPX_RUNNING) return;
    if ((GP.mute == 1 || (GP.mute == 2 && st->iter > 0)) && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) return;   // plays dead
    if (rs_abort_raised(st)) return;
    if (GP.mute == 3 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) rs_wait_for_abort(st);
    const int t = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    const int C = P.C, m = P.R - 1, rhsc = C - 1;
    const int rpw = (m + G - 1) / G;            // <= GP.rt (host): the first RPW of them live in registers, the rest in LDS
    const int row0 = w * rpw;
    const int nloc = max(0, min(rpw, m - row0));
    const int mp = (m + 1) & ~1;
    const int gld = P.ld;
    const int ld = rr_tile_width(C, KC);        // width of the register tile (a multiple of 2 and of KC): NT >= ld / KC, ld <= gld (host)
    double* obj = rr_lds;                       // [ld]
    double* prow = obj + ld;                    // [ld]
    double* rowbuf = prow + ld;                 // [ld]   the owner's pivot row as it stands (before the division)
    double* col = rowbuf + ld;                  // [mp]   gathered per-row values
    const int rt = GP.rt;                       // >= RPW
    double* fac = col + mp;                     // [rt+1]
    double* ca = fac + rt + 1;                  // [rt]   column qc of the local rows
    double* cr = ca + rt;                       // [rt]   RHS column of the local rows, kept current from round to round (as the tile's own)
    double* fn = cr + rt;                       // [rt]   column qc as the lookahead left it = the next pivot's factors when it enters
    // Local rows RPW .. nloc-1 live in LDS (the 120 KB the arrays above leave free: ~11 more rows of a config-4 node, so a node
    // takes 20 workgroups instead of 28 and TWELVE are in flight).  They are swept K0b-style, lane j mod NT on column j.
    double* lrows = fn + rt + 1;                // [nl][ld], 16-byte aligned like the arrays in front (ld, mp even; 4 rt + 2 doubles of per-row arrays)
    const int nl = max(0, nloc - RPW);

    const int jt = KC * t;                      // this lane's columns: jt .. jt + KC - 1
    const bool mine = jt < ld;
    double reg[RPW][KC];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
#pragma unroll
        for (int c = 0; c < KC; ++c) reg[i][c] = 0.0;
        if (i < nloc && mine) {
            const double* src = P.T + (size_t)(row0 + i) * gld + jt;
            if (KC == 2) { const double2 v = *reinterpret_cast<const double2*>(src); reg[i][0] = v.x; reg[i][KC - 1] = v.y; }
            else {
#pragma unroll
                for (int c = 0; c < KC; ++c) reg[i][c] = src[c];
            }
        }
    }
    for (int i = 0; i < nl; ++i) {
        const double* src = P.T + (size_t)(row0 + RPW + i) * gld;
        for (int j = 2 * t; j < ld; j += 2 * NT) *reinterpret_cast<double2*>(lrows + (size_t)i * ld + j) = *reinterpret_cast<const double2*>(src + j);
    }
    {
        const double* src = P.T + (size_t)m * gld;
        for (int j = 2 * t; j < ld; j += 2 * NT) {
            *reinterpret_cast<double2*>(obj + j) = *reinterpret_cast<const double2*>(src + j);
            *reinterpret_cast<double2*>(prow + j) = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    // column c of the local rows -> dst[0..RPW): the registers of the one lane that owns it
    auto column_out = [&](int c, double* dst) __attribute__((always_inline)) {
        if (t == c / KC) {
            // the lane's column by bit masks: written as `odd ? reg[i].y : reg[i].x` the compiler turns the select of two values into a
            // select of two ADDRESSES and moves the whole register tile into scratch memory (480 bytes per lane, 4.6x slower per step)
            const int cc = c - KC * t;
            long long mk[KC];
#pragma unroll
            for (int u = 0; u < KC; ++u) mk[u] = -(long long)(cc == u);
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                long long b = 0;
#pragma unroll
                for (int u = 0; u < KC; ++u) b |= __double_as_longlong(reg[i][u]) & mk[u];
                dst[i] = __longlong_as_double(b);
            }
        }
        if (t < nl) dst[RPW + t] = lrows[(size_t)t * ld + c];
    };

    int phase = P.dual ? st->phase : 2;
    int fdf_count = st->fdf_count, dual_iter = st->dual_iter, primal_count = st->primal_count, iter = st->iter;
    unsigned gen = *N.xgen;
    int status = LPX_RUNNING;
    bool hung = false;
    int r = -1, qlast = -1;
    int qc = (phase != 1) ? rr_first_min_below<NT>(obj, rhsc, eps, s_v, s_i) : -1;
    bool publish_now = GP.chunk > 0;
    // A column of the local rows costs its owner lane RPW masked moves and LDS stores (~0.3 us): the RHS column is therefore kept as
    // an LDS replica that the lookahead advances with the tile's own arithmetic, and the column the lookahead corrected for the next
    // ratio test IS the next pivot's factor column when that column enters (fn, valid for column fn_col)
    column_out(rhsc, cr);
    int fn_col = -1;

    // DEFER: the update of pivot k (factors `fac`, normalised row `prow`, the owner's row `pend_skip`) stays PENDING through the end of
    // round k and the decision of round k+1, and is applied where a workgroup used to sleep and poll: between the decision and the
    // arrival of pivot k+1's row (the owner of that row publishes first and updates then).  Whoever needs a value of the tile before
    // that forms it as the update would (mul, then sub): the owner the pivot row it extracts, the republish path its column -- the
    // lookahead has always worked that way.  On the way out (final status, end of the launch) the pending update is applied behind the loop.
    bool pend_upd = false; int pend_skip = -1;
    if (DEFER) { for (int i = t; i <= rt; i += NT) fac[i] = 0.0; __syncthreads(); }
    RR_T0
    for (int k = 0; k < GP.chunk; ++k) {
        int fail = 0, q = -1;
        if (publish_now) {
            if (phase == 0 && (fdf_count >= N.fdf_guard || qc < 0)) { phase = 1; qc = -1; }
            if (phase != 1 && qc >= 0) column_out(qc, ca);
            __syncthreads();
            if (t < nloc) {
                const double rhs0 = cr[t];
                double v = rhs0;
                if (phase != 1) {
                    double a0 = qc >= 0 ? ca[t] : 0.0;
                    if (DEFER && pend_upd && qc >= 0) {                         // column qc as the pending update will leave it
                        if (t == pend_skip) a0 = prow[qc];
                        else { const double prod = fac[t] * prow[qc]; a0 = a0 - prod; }
                    }
                    v = a0 > eps ? rhs0 / a0 : __builtin_inf();                 // :229-233
                }
                rs_publish(P.xr + 2 * ((size_t)((gen + 1u) & 1u) * P.mcap + row0 + t), v, gen + 1u);
            }
            publish_now = false;
        }
        ++gen;
        const int par = (int)(gen & 1u);
        // ---- exchange 1: one value per row ----------------------------------------------------------------------------
        for (int base = t; base < m; base += NT * 2) {
            double val[2]; const int i0 = base, i1 = base + NT; const int cnt = i1 < m ? 2 : 1;
            if (!rs_gather2(P.xr + 2 * (size_t)par * P.mcap, i0, i1, cnt, gen, val)) fail = 1;
            col[i0] = val[0];
            if (cnt > 1) col[i1] = val[1];
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RR_T(1);

        // ---- the decision of lpx_select_body, replicated (as lpx_resident_group) ------------------------------------------
        int final_status = LPX_RUNNING;
        bool republish = false;
        r = -1;
        for (int hop = 0; hop < 3 && final_status == LPX_RUNNING && r < 0 && !republish; ++hop) {
            if (phase == 1) {
                if (dual_iter >= N.max_iter) { final_status = LPX_ITER_LIMIT; break; }
                r = rr_first_min_below<NT>(col, m, eps, s_v, s_i);
                if (r < 0) {
                    if (N.cleanup) {
                        const int qe = rr_first_min_below<NT>(obj, rhsc, eps, s_v, s_i);
                        if (qe >= 0) { phase = 2; qc = qe; republish = true; break; }
                    }
                    final_status = LPX_OPTIMAL; break;
                }
                q = -2;
            } else {
                if (phase == 0 && fdf_count >= N.fdf_guard) { phase = 1; qc = -1; republish = true; break; }
                if (phase == 2 && primal_count >= N.max_iter - dual_iter) { final_status = LPX_ITER_LIMIT; break; }
                q = qc;
                if (q < 0) { if (phase == 0) { phase = 1; qc = -1; republish = true; break; } final_status = LPX_OPTIMAL; break; }
                r = rr_hysteresis(m, phase == 0 ? N.tol_fdf : N.tol_primal, col, s_v, s_i, &s_out);
                if (r < 0) { q = -1; if (phase == 0) { phase = 1; qc = -1; republish = true; break; } final_status = LPX_UNBOUNDED; break; }
            }
        }
        if (republish) { publish_now = true; continue; }
        if (final_status != LPX_RUNNING || r < 0) { status = (final_status == LPX_RUNNING) ? LPX_OPTIMAL : final_status; break; }

        RR_T(2);
        // ---- exchange 2: the owner normalises row r (and, in the dual loop, chooses the entering column) ------------------
        const int owner = r / rpw, rl = r - owner * rpw;
        u64* xp = P.xp + 2 * (size_t)par * (gld + 8);
        if (w == owner) {
            // the pivot row out of the registers: register `rl` of every lane (rl is uniform, the chain is unrolled)
            const bool fix = DEFER && pend_upd;                                 // the row as the pending update will leave it
            const bool was = fix && rl == pend_skip;                            // ... which made it the normalised row of the last pivot
            const double fp = fix ? fac[rl] : 0.0;
            if (rl >= RPW) {                   // an LDS row
                const double* src = lrows + (size_t)(rl - RPW) * ld;
                for (int j = 2 * t; j < ld; j += 2 * NT) {
                    double2 x = *reinterpret_cast<const double2*>(src + j);
                    if (fix) {
                        const double2 p2 = *reinterpret_cast<const double2*>(prow + j);
                        if (was) x = p2;
                        else { double prod = fp * p2.x; x.x = x.x - prod; prod = fp * p2.y; x.y = x.y - prod; }
                    }
                    *reinterpret_cast<double2*>(rowbuf + j) = x;
                }
            } else if (mine) {
                double v[KC];
#pragma unroll
                for (int c = 0; c < KC; ++c) v[c] = reg[0][c];
#pragma unroll
                for (int i = 1; i < RPW; ++i)
                    if (i == rl) {
#pragma unroll
                        for (int c = 0; c < KC; ++c) v[c] = reg[i][c];
                    }
                if (fix) {
#pragma unroll
                    for (int c = 0; c < KC; ++c) {
                        const double pc = prow[jt + c];
                        if (was) v[c] = pc; else { const double prod = fp * pc; v[c] = v[c] - prod; }
                    }
                }
#pragma unroll
                for (int c = 0; c < KC; ++c) rowbuf[jt + c] = v[c];
            }
            rs_barrier_lds();                   // LDS only: a full barrier would also wait for the lookahead's write-through stores
            if (phase == 1) {                                                   // entering column of the dual loop, :79-91
                if ((t >> 6) == 0) {
                    const int win = wave_hysteresis_argmin<DualColRatio, true, 4>(rhsc, N.tol_dual, DualColRatio{rowbuf, obj, eps});
                    if (t == 0) s_out = win;
                }
                __syncthreads();
                q = s_out;
            }
            if (q >= 0) {
                const double piv = rowbuf[q];
                if (DEFER) rs_barrier_lds();                                    // rowbuf[q] is overwritten below: every lane has its copy first
                for (int j = t; j < C; j += NT) {
                    const double p = rowbuf[j] / piv;                           // true division, :250
                    rs_publish(xp + 2 * (size_t)j, p, gen);
                    if (DEFER) rowbuf[j] = p; else prow[j] = p;                 // DEFER: prow still holds the pending update's row
                }
            }
            if (phase == 1) {
                __syncthreads();
                if (t == 0) rs_publish(xp + 2 * (size_t)gld, (double)q, gen);   // header {q} behind the row
            }
        }
        if (DEFER) {
            const bool had = pend_upd;
            RR_UPDATE(pend_skip); pend_upd = false; pend_skip = -1;    // unconditional: before the first pivot fac is zero (x - 0 * p = x)
            rs_barrier_lds();                   // prow may be rewritten now
            if (w == owner) { if (q >= 0) for (int j = t; j < C; j += NT) prow[j] = rowbuf[j]; }
            else if (!had) { __builtin_amdgcn_s_sleep(15); for (int z = 0; z < C; z += 1024) __builtin_amdgcn_s_sleep(5); if (phase == 1) __builtin_amdgcn_s_sleep(25); }
        }
        if (w == owner) {
        } else if (phase != 1) {
            if (!DEFER) { __builtin_amdgcn_s_sleep(15); for (int z = 0; z < C; z += 1024) __builtin_amdgcn_s_sleep(5); }
            bool first = true;
            for (int base = t; base < C; base += NT * KC) {
                double val[KC]; const int cnt = min(KC, (C - base + NT - 1) / NT);
                unsigned pend = (1u << cnt) - 1u;
                if (first && !rr_gather<KC, NT>(xp, base, cnt, gen, val, 1u, &pend)) {
                    if (!rs_wait(xp + 2 * (size_t)(C - 1), gen)) fail = 1;
                }
                first = false;
                if (pend && !rr_gather<KC, NT>(xp, base, cnt, gen, val, RS_SPIN_MAX, &pend)) fail = 1;
#pragma unroll
                for (int u = 0; u < KC; ++u) if (u < cnt) prow[base + u * NT] = val[u];
            }
        } else {
            if (!DEFER) {
                __builtin_amdgcn_s_sleep(15);
                for (int z = 0; z < C; z += 1024) __builtin_amdgcn_s_sleep(5);
                __builtin_amdgcn_s_sleep(25);                                   // the owner scans its row first
            }
            // the header first (one granule, the same address in every lane), then the row
            {
                double hval[2];
                if (!rs_gather2(xp, gld, gld, 1, gen, hval)) fail = 1;
                q = fail ? -1 : (int)hval[0];
            }
            if (q >= 0)
                for (int base = t; base < C; base += NT * KC) {
                    double val[KC]; const int cnt = min(KC, (C - base + NT - 1) / NT);
                    if (!rr_gather<KC, NT>(xp, base, cnt, gen, val)) fail = 1;
#pragma unroll
                    for (int u = 0; u < KC; ++u) if (u < cnt) prow[base + u * NT] = val[u];
                }
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        if (q < 0) { r = -1; status = LPX_INFEASIBLE; break; }                  // :92-96 (dual loop only)
        RR_T(3);

        // ---- column factors of this pivot, objective replica, next entering column ------------------------------------------
        if (fn_col == q) { if (t < rt) fac[t] = fn[t]; }                        // the lookahead of the last round already formed this column
        else column_out(q, fac);
        if (t == NT - 1) fac[rt] = obj[q];
        rs_barrier_lds();
        const double fobj = fac[rt];
        const int skip = (w == owner) ? rl : -1;
        MinIdx best; best.v = -eps; best.i = INT_MAX;
        if (t < RS_RT) {
            for (int j = 2 * t; j < ld; j += 2 * RS_RT) {
                const double2 p = *reinterpret_cast<const double2*>(prow + j);
                double2 o = *reinterpret_cast<double2*>(obj + j);
                double prod = fobj * p.x; o.x = o.x - prod;
                prod = fobj * p.y; o.y = o.y - prod;
                *reinterpret_cast<double2*>(obj + j) = o;
                if (j < rhsc && o.x < best.v) { best.v = o.x; best.i = j; }
                if (j + 1 < rhsc && o.y < best.v) { best.v = o.y; best.i = j + 1; }
            }
        }
        best = first4_min_idx(best, s_v, s_i);
        if (w == 0 && t == 0) {
            N.basis[r] = q;                                                     // basis[leaving] = entering, :110
            if (iter < N.trace_cap) { N.trace[2 * iter] = r; N.trace[2 * iter + 1] = q; }
        }
        qlast = q;
        ++iter;
        if (phase == 0) ++fdf_count; else if (phase == 1) ++dual_iter; else ++primal_count;
        qc = (phase != 1) ? (best.i == INT_MAX ? -1 : best.i) : -1;
        RR_T(4);
        // ---- lookahead: next round's per-row values leave before the bulk of the update ------------------------------------
        const bool look = k + 1 < GP.chunk;
        if (look) {
            if (phase == 0 && (fdf_count >= N.fdf_guard || qc < 0)) { phase = 1; qc = -1; }
            if (qc >= 0) column_out(qc, ca);                                    // column qc as it stands BEFORE this pivot's update
        }
        rs_barrier_lds();
        fn_col = -1;
        if (look) {
            if (t < nloc) {
                double a = 0.0, rhs;
                if (t == skip) { if (qc >= 0) a = prow[qc]; rhs = prow[rhsc]; }
                else {
                    const double f = fac[t];
                    if (qc >= 0) { const double prod = f * prow[qc]; a = ca[t] - prod; }
                    const double prod2 = f * prow[rhsc]; rhs = cr[t] - prod2;
                }
                cr[t] = rhs;                                                    // the RHS replica moves on with the tile
                if (qc >= 0) fn[t] = a;
                const double v = (phase == 1) ? rhs : (a > eps ? rhs / a : __builtin_inf());
                rs_publish(P.xr + 2 * ((size_t)(par ^ 1) * P.mcap + row0 + t), v, gen + 1u);
            }
            if (qc >= 0) fn_col = qc;
        }
        RR_T(0);
        if (DEFER) { pend_upd = true; pend_skip = skip; }    // applied in the next round (or on the way out)
        else RR_UPDATE(skip);
        rs_barrier_lds();                       // fac / ca / cr / prow / rowbuf are rewritten by the next round
        RR_T(5);
    }

    if (hung) {
        if (t == 0) { atomicOr(&st->pad[1], 1); if (N.st_host) N.st_host->pad[1] = 1; }
        return;
    }
    if (DEFER && pend_upd) RR_UPDATE(pend_skip);                                 // the last pivot's update, on the way out
    if (mine) {
#pragma unroll
        for (int i = 0; i < RPW; ++i)
            if (i < nloc) {
                double* dst = P.T + (size_t)(row0 + i) * gld + jt;
                if (KC == 2) *reinterpret_cast<double2*>(dst) = make_double2(reg[i][0], reg[i][KC - 1]);
                else {
#pragma unroll
                    for (int c = 0; c < KC; ++c) dst[c] = reg[i][c];
                }
            }
    }
    for (int i = 0; i < nl; ++i) {
        double* dst = P.T + (size_t)(row0 + RPW + i) * gld;
        for (int j = 2 * t; j < ld; j += 2 * NT) *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(lrows + (size_t)i * ld + j);
    }
    if (w == 0) {
        double* dst = P.T + (size_t)m * gld;
        for (int j = 2 * t; j < ld; j += 2 * NT)
            *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(obj + j);
        if (t == 0) {
            st->status = status; st->iter = iter; st->phase = phase;
            st->fdf_count = fdf_count; st->dual_iter = dual_iter; st->primal_count = primal_count;
            st->r = status == LPX_RUNNING ? r : -1; st->q = status == LPX_RUNNING ? qlast : -1;
            if (DevState* hm = N.st_host) {
                hm->status = status; hm->iter = iter; hm->phase = phase;
                hm->fdf_count = fdf_count; hm->dual_iter = dual_iter; hm->primal_count = primal_count;
                hm->r = status == LPX_RUNNING ? r : -1; hm->q = status == LPX_RUNNING ? qlast : -1;
            }
            *N.xgen = gen;
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
// shapes the register kernel is instantiated for: (lanes, rows per workgroup, columns per lane)
// (register rows; the LDS takes ~12 more rows of a 1284-wide node, ~15 of a 1024-wide one: resident_regs_shape)
static constexpr int RR_NT_A = 768, RR_RPW_A = 22;     // tableaux up to 1536 columns, two columns per lane (r03's first form)
static constexpr int RR_NT_B = 512, RR_RPW_B = 40;     // up to 1024 columns (48 rows spilled 74 registers once the LDS rows were there; 40: 3)
static constexpr int RR_NT_C = 512, RR_RPW_C = 26;     // up to 1536 columns, THREE columns per lane: 26 + 11 rows -> 21 workgroups per config-4 node -> twelve nodes

static int rr_kc(int cfg) { return cfg == 3 ? 3 : 2; }
static int rr_reg_rows(int cfg) { return cfg == 1 ? RR_RPW_B : (cfg == 2 ? RR_RPW_A : RR_RPW_C); }
// Dynamic LDS a workgroup may ask for: the CU's 160 KB less the kernels' static records (measured by resident_regs_init; 464 B today).
// 158 KB held one LDS row less for nodes of 1285+ columns (levels 3+ of config 4): 11 nodes on chip instead of 12.
static size_t g_rr_lds_budget = (size_t)158 * 1024;
size_t resident_regs_lds_budget() { return g_rr_lds_budget; }

// rt = rows a workgroup may hold (the per-row arrays are sized for it); the rows beyond the configuration's register rows live in LDS
size_t resident_regs_lds(int R, int C, int rt, int cfg)
{
    const int m = R - 1, ld = rr_tile_width(C, rr_kc(cfg));
    const int lrows = std::max(0, rt - rr_reg_rows(cfg));
    return sizeof(double) * ((size_t)3 * ld + (size_t)((m + 1) & ~1) + (size_t)4 * rt + 2 + (size_t)lrows * ld);
}

// 0 = this group does not fit the register kernel; else the configuration (1 = B, 2 = A, 3 = C), with *rpw_max the rows a
// workgroup can hold: the configuration's register rows plus what the LDS takes beside the row buffers (a node of mmax + 1 rows).
// min_ld = the smallest row pitch of the group: the tile must not be wider than a row in memory.
int resident_regs_shape(int maxC, int min_ld, int mmax, int* rpw_max)
{
    static const int wide = [] { const char* e = std::getenv("LPX_RESIDENT_REGS_KC"); return e ? std::atoi(e) : 3; }();   // diagnostic: 2 = the two-column form for wide nodes
    static const bool lds_rows = [] { const char* e = std::getenv("LPX_RESIDENT_REGS_LDS"); return !(e && e[0] == '0'); }();   // diagnostic: 0 = register rows only
    int cfg = 0;
    if (rr_tile_width(maxC, 2) <= 2 * RR_NT_B && rr_tile_width(maxC, 2) <= min_ld) cfg = 1;
    else if (wide == 3 && rr_tile_width(maxC, 3) <= 3 * RR_NT_C && rr_tile_width(maxC, 3) <= min_ld) cfg = 3;
    else if (rr_tile_width(maxC, 2) <= 2 * RR_NT_A && rr_tile_width(maxC, 2) <= min_ld) cfg = 2;
    if (!cfg) return 0;
    const int regs = rr_reg_rows(cfg), ld = rr_tile_width(maxC, rr_kc(cfg));
    const long long fixed = 3LL * ld + ((mmax + 1) & ~1) + 4LL * regs + 2;
    const long long left = (long long)(g_rr_lds_budget / sizeof(double)) - fixed;
    *rpw_max = regs + (lds_rows && left > 0 ? (int)(left / (ld + 4)) : 0);
    return cfg;
}

hipError_t resident_regs_init()
{
    const void* ks[4] = {reinterpret_cast<const void*>(lpx_resident_group_r<RR_NT_A, RR_RPW_A, 3, 2, false>),
                         reinterpret_cast<const void*>(lpx_resident_group_r<RR_NT_B, RR_RPW_B, 2, 2, false>),
                         reinterpret_cast<const void*>(lpx_resident_group_r<RR_NT_C, RR_RPW_C, 2, 3, true>),
                         reinterpret_cast<const void*>(lpx_resident_group_r<RR_NT_C, RR_RPW_C, 2, 3, false>)};
    size_t stat = 512;
    for (const void* k : ks) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, k) == hipSuccess) stat = std::max(stat, (size_t)a.sharedSizeBytes);
        else (void)hipGetLastError();
    }
    g_rr_lds_budget = ((size_t)160 * 1024 - stat) & ~(size_t)15;
    hipError_t e = hipSuccess;
    for (const void* k : ks)
        if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_rr_lds_budget);
    return e;
}

hipError_t launch_resident_regs(const void* nodes_dev, int nodes, int grid, int cfg, int rt, size_t lds, int chunk, hipStream_t s)
{
    ResGroupParamsR p; p.nodes = static_cast<const ResNode*>(nodes_dev); p.chunk = chunk; p.rt = std::max(rt, rr_reg_rows(cfg));
    static const int mute = [] { const char* e = std::getenv("LPX_RESIDENT_TEST_MUTE"); return e ? std::atoi(e) : 0; }();
    p.mute = mute;
    if (cfg == 2) hipLaunchKernelGGL((lpx_resident_group_r<RR_NT_A, RR_RPW_A, 3, 2, false>), dim3(grid, nodes), dim3(RR_NT_A), lds, s, p);
    else if (cfg == 3) {
        static const bool defer = [] { const char* e = std::getenv("LPX_RESIDENT_REGS_DEFER"); return !(e && e[0] == '0'); }();   // diagnostic: 0 = update at the end of its own round
        if (defer) hipLaunchKernelGGL((lpx_resident_group_r<RR_NT_C, RR_RPW_C, 2, 3, true>), dim3(grid, nodes), dim3(RR_NT_C), lds, s, p);
        else hipLaunchKernelGGL((lpx_resident_group_r<RR_NT_C, RR_RPW_C, 2, 3, false>), dim3(grid, nodes), dim3(RR_NT_C), lds, s, p);
    }
    else hipLaunchKernelGGL((lpx_resident_group_r<RR_NT_B, RR_RPW_B, 2, 2, false>), dim3(grid, nodes), dim3(RR_NT_B), lds, s, p);
    return hipGetLastError();
}

}  // namespace lpx
