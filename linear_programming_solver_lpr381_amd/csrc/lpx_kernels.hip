// lpx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the dense-tableau simplex loop.
//
// Built with -ffp-contract=off: `t - f*p` must round twice (v_mul_f64 + v_add_f64), exactly as the
// reference's scalar C# does (Models/PrimalSimplex.cs:255); pivot-row normalisation uses true IEEE
// division (Models/PrimalSimplex.cs:250), never a reciprocal multiply.
//
// Two launches per pivot on one stream:
//   lpx_select  (1 workgroup x 1024 lanes)  ChooseEntering + ChooseLeaving + pivot prep
//   lpx_update  (>> 256 workgroups)         rank-1 update of the whole tableau, HBM-bound
#include <cstdlib>
#include "lpx_resident.h"      // rs_hysteresis: the hysteresis scan over ratios held in LDS (also pulls in lpx_block.h)
#include <hip/hip_ext.h>

namespace lpx {

// ------------------------------------------------------------------------------------------------
// lpx_select: one workgroup decides the next pivot and prepares the update's operands.
// ------------------------------------------------------------------------------------------------
// The two ratio scans of the dual path as real calls: inlined three times (one per phase) their 16-ratio register blocks pushed
// the 1024-lane kernel (128 VGPRs per lane) into scratch spills -- 13 in lpx_select, 22 in lpx_select_b (code-object metadata).
__device__ __attribute__((noinline)) int sel_row_scan(int m, double tol, const double* col, size_t cs, const double* rhsb, double eps, int* s_out)
{
    return block_hysteresis_auto<SEL_NW>(m, tol, RowRatio{col, cs, rhsb, 1, eps}, s_out);
}
__device__ __attribute__((noinline)) int sel_col_scan(int L, double tol, const double* lrow, const double* zrow, double eps, int* s_out)
{
    return block_hysteresis_auto<SEL_NW>(L, tol, DualColRatio{lrow, zrow, eps}, s_out);
}

// `ratios`: dynamic LDS for max(m, C-1) doubles, or nullptr when the tableau is too long for it (then the scans read their
// operands from global memory on one wave, as in round 1).  With it, ALL 1024 lanes gather the operands and divide in
// parallel (one round trip, ~1 division per lane), and the chain runs over LDS: minimum + band count on four waves, exact
// replay on wave 0 only when rows tie (rs_hysteresis, lpx_resident.h -- the scan the resident kernels use, bit-identical).
__device__ __forceinline__ void lpx_select_body(const SelParams& P, double* ratios)
{
    __shared__ int s_out;
    __shared__ double s_v[SEL_NW];
    __shared__ int s_i[SEL_NW];

    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;              // uniform: loop already finished

    const int t = threadIdx.x;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    const int m = R - 1;
    const int rhs = C - 1;
    const size_t ld = (size_t)P.ld;
    double* T = P.T;
    // contiguous copy of the RHS column: filled once per run by lpx_rhs_init, kept current by lpx_update (rows i != r)
    // and by this kernel (row r) -- the dual loop's leaving-row scan and every ratio test read it instead of a strided column
    const double* rhsb = P.rhsbuf;

    int phase = st->phase;
    const int fdf_count = st->fdf_count, dual_iter = st->dual_iter, primal_count = st->primal_count;
    const int iter = st->iter;
    int r = -1, q = -1;
    int final_status = LPX_RUNNING;

    {
        // state machine: ForceDualFeasibility -> dual loop -> (repaired mode) primal clean-up
        for (int hop = 0; hop < 3 && final_status == LPX_RUNNING && r < 0; ++hop) {
            if (phase == 0) {
                // ForceDualFeasibility, Models/DualSimplex.cs:195-228
                if (fdf_count >= P.fdf_guard) { phase = 1; continue; }
                q = block_first_min_below(T + (size_t)m * ld, 1, rhs, P.eps, s_v, s_i);
                if (q < 0) { phase = 1; continue; }
                if (ratios) {
                    for (int i = t; i < m; i += SEL_NT) { const double a = T[(size_t)i * ld + q]; ratios[i] = a > P.eps ? rhsb[i] / a : __builtin_inf(); }
                    __syncthreads();
                    r = rs_hysteresis(m, P.tol_fdf, ratios, s_v, s_i, &s_out);
                } else
                r = sel_row_scan(m, P.tol_fdf, T + q, ld, rhsb, P.eps, &s_out);
                if (r < 0) { q = -1; phase = 1; continue; }
            } else if (phase == 1) {
                // dual loop, Models/DualSimplex.cs:36-113
                if (dual_iter >= P.max_iter) { final_status = LPX_ITER_LIMIT; break; }
                r = block_first_min_below(rhsb, 1, m, P.eps, s_v, s_i);
                if (r < 0) {
                    if (P.cleanup) {
                        int qe = block_first_min_below(T + (size_t)m * ld, 1, rhs, P.eps, s_v, s_i);
                        if (qe >= 0) { phase = 2; continue; }
                    }
                    final_status = LPX_OPTIMAL; break;
                }
                if (ratios) {
                    const double* lrow = T + (size_t)r * ld; const double* zrow = T + (size_t)m * ld;
                    for (int j = t; j < rhs; j += SEL_NT) { const double a = lrow[j]; ratios[j] = a < -P.eps ? zrow[j] / (-a) : __builtin_inf(); }
                    __syncthreads();
                    q = rs_hysteresis(rhs, P.tol_dual, ratios, s_v, s_i, &s_out);
                } else
                q = sel_col_scan(rhs, P.tol_dual, T + (size_t)r * ld, T + (size_t)m * ld, P.eps, &s_out);
                if (q < 0) { r = -1; final_status = LPX_INFEASIBLE; break; }
            } else {
                // primal loop, Models/PrimalSimplex.cs:92-124
                if (primal_count >= P.max_iter - dual_iter) { final_status = LPX_ITER_LIMIT; break; }
                q = block_first_min_below(T + (size_t)m * ld, 1, rhs, P.eps, s_v, s_i);
                if (q < 0) { final_status = LPX_OPTIMAL; break; }
                if (ratios) {
                    for (int i = t; i < m; i += SEL_NT) { const double a = T[(size_t)i * ld + q]; ratios[i] = a > P.eps ? rhsb[i] / a : __builtin_inf(); }
                    __syncthreads();
                    r = rs_hysteresis(m, P.tol_primal, ratios, s_v, s_i, &s_out);
                } else
                r = sel_row_scan(m, P.tol_primal, T + q, ld, rhsb, P.eps, &s_out);
                if (r < 0) { q = -1; final_status = LPX_UNBOUNDED; break; }
            }
        }
    }

    if (final_status != LPX_RUNNING || r < 0) {
        if (t == 0) {
            st->status = (final_status == LPX_RUNNING) ? LPX_OPTIMAL : final_status;
            st->phase = phase; st->r = -1; st->q = -1;
        }
        return;
    }

    // ---- pivot prep (Models/PrimalSimplex.cs:249-250, :254): snapshot the pivot column, normalise
    // the pivot row in place and into `prow` so the update kernel never reads what it overwrites.
    const double piv = T[(size_t)r * ld + q];          // one address for the whole workgroup: a broadcast load
    for (int i = t; i < R; i += SEL_NT)
        P.pcol[i] = (i == r) ? 0.0 : T[(size_t)i * ld + q];
    __syncthreads();                                   // pivot and column read before the row is rewritten
    double* trow = T + (size_t)r * ld;
    for (int j = t; j < C; j += SEL_NT) {
        double p = trow[j] / piv;
        trow[j] = p;
        P.prow[j] = p;
        if (j == rhs) P.rhsbuf[r] = p;                 // lpx_update leaves row r alone
    }
    if (t == 0) {
        P.basis[r] = q;                                // basis[leaving] = entering, :110
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        st->iter = iter + 1;
        st->r = r; st->q = q; st->phase = phase; st->qn = -1;
        if (phase == 0) st->fdf_count = fdf_count + 1;
        else if (phase == 1) st->dual_iter = dual_iter + 1;
        else st->primal_count = primal_count + 1;
    }
}

// contiguous copy of the RHS column, once at the start of a dual run (single and batched)
__device__ __forceinline__ void lpx_rhs_init_body(const SelParams& P)
{
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    for (int i = threadIdx.x; i < R; i += SEL_NT) P.rhsbuf[i] = P.T[(size_t)i * P.ld + (C - 1)];
}

// ------------------------------------------------------------------------------------------------
// Lookahead select (primal and forced-pivot paths).
//
// The gather-based select above spends most of its time in 64-cache-lines-per-instruction strided
// column reads on ONE CU.  Here the update kernel of pivot k writes, as by-products of the stream it
// already does, the column that pivot k+1 will enter on (`coln`) and the RHS column (`rhsbuf`), both
// contiguous.  That is possible because the entering column of pivot k+1 depends only on the
// objective row AFTER pivot k, which select(k) can compute itself (obj - f_m * prow, the very
// arithmetic the update kernel will repeat bit for bit).  So select(k):
//   ratio test on contiguous colc/rhsbuf  ->  r
//   normalise row r (contiguous)          ->  prow, T[r,:]
//   updated scan row u = T[s,:] - colc[s]*prow  ->  next entering column qn   (s = objective row;
//   forced mode: s = next forced row, rule = first |u| >= thresh from the next forced column)
// Column buffers ping-pong on pivot parity: update(k) reads colc as factors while writing coln.
// ------------------------------------------------------------------------------------------------
struct ScanRule { int forced; double eps; double thresh; int c0; int C; };

#ifdef LPX_STAMPS
// Diagnostic build only (never shipped): thread 0 accumulates s_memtime deltas per segment into ws[].
#define LPX_STAMP(slot)                                                                        \
    do { if (threadIdx.x == 0) { unsigned long long now_ = __builtin_amdgcn_s_memtime();       \
         reinterpret_cast<unsigned long long*>(P.ws)[(slot)] += now_ - stamp_prev_; stamp_prev_ = now_; } } while (0)
#define LPX_STAMP_BEGIN unsigned long long stamp_prev_ = __builtin_amdgcn_s_memtime(); \
    unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime();
#define LPX_STAMP_END do { if (threadIdx.x == 0) { reinterpret_cast<unsigned long long*>(P.ws)[14] += __builtin_amdgcn_s_memrealtime() - rt0_; \
    reinterpret_cast<unsigned long long*>(P.ws)[15] += 1; } } while (0)
#define LPX_STAMP_MB(slot)                                                                     \
    do { if (threadIdx.x == 0 && blockIdx.x == 1) { unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
         reinterpret_cast<unsigned long long*>(P.part_v + 128)[(slot)] += now_ - stamp_prev_; stamp_prev_ = now_; } } while (0)
#define LPX_STAMP_END_MB do { if (threadIdx.x == 0 && blockIdx.x == 1) { reinterpret_cast<unsigned long long*>(P.part_v + 128)[14] += __builtin_amdgcn_s_memrealtime() - rt0_; \
    reinterpret_cast<unsigned long long*>(P.part_v + 128)[15] += 1; } } while (0)
#else
#define LPX_STAMP_MB(slot) do {} while (0)
#define LPX_STAMP_END_MB do {} while (0)
#define LPX_STAMP(slot) do {} while (0)
#define LPX_STAMP_BEGIN
#define LPX_STAMP_END do {} while (0)
#endif

__device__ __forceinline__ void rule_init(const ScanRule& R, MinIdx& m)
{
    m.v = R.forced ? 0.0 : -R.eps; m.i = INT_MAX;
}
__device__ __forceinline__ void rule_feed(const ScanRule& R, MinIdx& m, int j, double u)
{
    if (R.forced) {
        if (fabs(u) >= R.thresh) { int off = j - R.c0; if (off < 0) off += R.C; if (off < m.i) m.i = off; }
    } else {
        if (j < R.C - 1 && u < m.v) { m.v = u; m.i = j; }      // ChooseEntering, :205-220
    }
}
__device__ __forceinline__ int rule_decode(const ScanRule& R, const MinIdx& m)
{
    if (m.i == INT_MAX) return -1;
    if (!R.forced) return m.i;
    int q = R.c0 + m.i; if (q >= R.C) q -= R.C;
    return q;
}

// Slow path (once per solve, or after a skipped forced pivot): pick the next column from T as it
// stands and gather it plus the RHS column with strided reads.
template <int NT = SEL_NT>
__device__ int la_prepare_from_T(const SelParams& P, int R, int C, double* buf, int scanrow, const ScanRule& rule,
                                 double* s_v, int* s_i)
{
    const size_t ld = (size_t)P.ld;
    int qn = -1;
    if (scanrow >= 0) {
        MinIdx b; rule_init(rule, b);
        const double* srow = P.T + (size_t)scanrow * ld;
        for (int j = threadIdx.x; j < C; j += NT) rule_feed(rule, b, j, srow[j]);
        b = block_min_idx<NT>(b, s_v, s_i);
        qn = rule_decode(rule, b);
    }
    for (int i = threadIdx.x; i < R; i += NT) {
        if (qn >= 0) buf[i] = P.T[(size_t)i * ld + qn];
        P.rhsbuf[i] = P.T[(size_t)i * ld + (C - 1)];
    }
    return qn;
}

__device__ __forceinline__ void lpx_la_init_body(const SelParams& P)
{
    __shared__ double s_v[SEL_NW];
    __shared__ int s_i[SEL_NW];
    DevState* st = P.st;
    if (st->status != LPX_RUNNING) return;
    const int iter = st->iter;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    double* colc = (iter & 1) ? P.col1 : P.col0;
    ScanRule rule; rule.forced = (P.mode == MODE_FORCED); rule.eps = P.eps; rule.thresh = P.fthresh;
    rule.C = C; rule.c0 = 0;
    int scanrow = R - 1;
    if (rule.forced) {
        const int k = st->forced_k;
        scanrow = k < P.fcount ? P.frows[k] : -1;
        rule.c0 = k < P.fcount ? P.fcols[k] : 0;
    }
    int qn = la_prepare_from_T(P, R, C, colc, scanrow, rule, s_v, s_i);
    if (threadIdx.x == 0) {
        st->qn = qn;
        if (P.us) { *P.us = *st; P.us->qn = qn; P.part_i[MB_CNT] = 0; }
    }
}

__global__ __launch_bounds__(SEL_NT) void lpx_select_la(SelParams P)
{
    __shared__ int s_out;
    __shared__ double s_v[SEL_NW];
    __shared__ int s_i[SEL_NW];

    DevState* st = P.st;
    LPX_STAMP_BEGIN
    if (st->status != LPX_RUNNING) return;
    LPX_STAMP(0);

    const int t = threadIdx.x;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    const int m = R - 1;
    const size_t ld = (size_t)P.ld;
    double* T = P.T;
    const int iter = st->iter;
    const int primal_count = st->primal_count;
    double* colc = (iter & 1) ? P.col1 : P.col0;     // column q of the current tableau
    double* coln = (iter & 1) ? P.col0 : P.col1;     // receives column qn of the next one
    const int q = st->qn;
    int r = -1, scanrow = -1;
    int final_status = LPX_RUNNING;
    ScanRule rule; rule.forced = (P.mode == MODE_FORCED); rule.eps = P.eps; rule.thresh = P.fthresh;
    rule.C = C; rule.c0 = 0;

    if (rule.forced) {
        const int k = st->forced_k;
        if (k >= P.fcount) {
            final_status = LPX_OPTIMAL;
        } else {
            r = P.frows[k];
            scanrow = (k + 1 < P.fcount) ? P.frows[k + 1] : -1;
            rule.c0 = (k + 1 < P.fcount) ? P.fcols[k + 1] : 0;
            if (t == 0) { P.fchosen[k] = q; st->forced_k = k + 1; }
            if (q < 0) {                                  // no eligible column: skip this pivot
                int qn = la_prepare_from_T(P, R, C, colc, scanrow, rule, s_v, s_i);
                if (t == 0) { st->r = -1; st->q = -1; st->qn = qn; }
                return;
            }
        }
    } else {
        // primal loop head, Models/PrimalSimplex.cs:95-106
        if (primal_count >= P.max_iter) final_status = LPX_ITER_LIMIT;
        else if (q < 0) final_status = LPX_OPTIMAL;
        else {
            r = block_hysteresis_argmin(m, P.tol_primal, RowRatio{colc, 1, P.rhsbuf, 1, P.eps}, &s_out);
            if (r < 0) final_status = LPX_UNBOUNDED;
            scanrow = m;
        }
    }
    LPX_STAMP(1);
    if (final_status != LPX_RUNNING) {
        if (t == 0) { st->status = final_status; st->r = -1; st->q = -1; }
        return;
    }

    // Pivot prep (Models/PrimalSimplex.cs:249-250) fused with the lookahead scan of the updated row.
    const double piv = colc[r];
    const bool same = (scanrow == r);
    const double fs = (scanrow >= 0 && !same) ? colc[scanrow] : 0.0;
    double* trow = T + (size_t)r * ld;
    const double* srow = T + (size_t)(scanrow >= 0 ? scanrow : 0) * ld;
    MinIdx best; rule_init(rule, best);
    for (int j = t; j < C; j += SEL_NT) {
        const double p = trow[j] / piv;
        trow[j] = p;
        P.prow[j] = p;
        if (scanrow >= 0) {
            const double u = same ? p : srow[j] - fs * p;     // what lpx_update will store at T[s,j]
            rule_feed(rule, best, j, u);
        }
    }
    LPX_STAMP(2);
    best = block_min_idx(best, s_v, s_i);                     // barrier inside: prow is complete
    LPX_STAMP(3);
    const int qn = (scanrow >= 0) ? rule_decode(rule, best) : -1;
    if (t == 0) {
        if (qn >= 0) coln[r] = P.prow[qn];                    // row r is not touched by lpx_update
        P.rhsbuf[r] = P.prow[C - 1];
        if (!rule.forced) P.basis[r] = q;                     // basis[leaving] = entering, :110
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        st->iter = iter + 1;
        st->r = r; st->q = q; st->qn = qn;
        if (!rule.forced) st->primal_count = primal_count + 1;
    }
    LPX_STAMP(4);
    LPX_STAMP_END;
}

// ------------------------------------------------------------------------------------------------
// lpx_update: T[i, :] -= pcol[i] * prow[:] for every row i != r  (Models/PrimalSimplex.cs:251-256,
// Models/DualSimplex.cs:240-245).  HBM-bound: 16 bytes of traffic per element, 2 flop.
//
// Work unit = one wave x (128 columns x UPD_ROWS rows): each lane owns two adjacent doubles
// (one 16-byte global_load_dwordx4 / global_store_dwordx4 per row, 1 KiB contiguous per wave per
// row), keeps its slice of the normalised pivot row in registers, takes the row factor from a
// scalar load, and has UPD_ROWS independent loads in flight.  Units are flattened over
// (row block, column chunk) so ragged widths waste at most part of one wave per row block.
// ------------------------------------------------------------------------------------------------
static constexpr int UPD_NT = 256;
static constexpr int UPD_ROWS = 8;
// Streaming variant for tableaux that cannot live in the 256 MiB Infinity Cache: one wave per workgroup, 3 rows per
// wave, non-temporal loads AND stores (`nt`: the lines are not kept in L2 / MALL, where they would only evict each
// other before the next pivot comes round).  Measured on 4097 x 12289 (403 MB), tools/kbench/store_variants.hip:
// 8 rows x 256-lane workgroups, default policy 141.7 us (5.69 TB/s); 3 rows x 64 lanes with nt on both sides 126.9 us
// (6.35 TB/s); nt on one side only, or nt with the 8-row tile, gains nothing.  Below ~1.2x the cache size the default
// policy wins (4096 x 8192 = 256 MiB: 77.7 us vs 83-88 us), so the launcher switches on the tableau's size.
static constexpr int UPDS_NT = 64;
static constexpr int UPDS_ROWS = 3;
static constexpr size_t UPD_STREAM_BYTES = (size_t)292 << 20;   // 306 MB: measured crossover (282 / 298 MB: this kernel wins, 315 MB: the mixed form)
// UPDM: above the cache size, storing ONE of the wave's three rows with the default policy (the other two and all loads
// nontemporal) is worth 5-8 %: that row is written through the Infinity Cache and found there by the next pivot's loads --
// as long as what is kept amounts to about one cache-full.  Measured on the product (tools/probe_policy.py, HIP events) and
// with every store's policy read off the ISA (tools/kbench/sweep_dir.hip, profiles/r02_kbench_sweep_dir.txt):
//   403 / 576 / 784 MB, all nt 126 / 185 / 249 us, LAST row default 116 / 167 / 228 us (first row: 119 / 172 / 229);
//   two rows default: 116 us at 403 MB, 195-202 us at 576 MB (it no longer fits), all default 140 us;
//   1074 / 1441 MB: one row in three no longer fits (347 / 499 us vs 342 / 462 all nt); the same row in every SECOND row
//   block (a sixth of the tableau) 330 / 447 us.
// Hence: one row in three of every `mixmod`-th row block, mixmod = ceil(bytes / 768 MiB); all-nt beyond 8 GiB (unmeasured).
static constexpr size_t UPD_MIXED_BYTES = (size_t)8192 << 20;
static constexpr size_t UPD_MIX_STEP_BYTES = (size_t)768 << 20;
static constexpr size_t FUSED_CACHED_BYTES = (size_t)152 << 20;   // fused (out-of-place) forms: both buffers at home in the Infinity Cache, see fused_policy

typedef double lpx_d2 __attribute__((ext_vector_type(2)));
// __builtin_nontemporal_load / _store lower to global_load_dwordx4 / global_store_dwordx4 ... nt on gfx950 and stay inside
// hipcc's s_waitcnt bookkeeping (an inline-asm load would not: cdna_hip_programming.md 5.7).
template <bool STREAM> __device__ __forceinline__ double2 upd_load(const double* p)
{
    if constexpr (STREAM) {
        const lpx_d2 v = __builtin_nontemporal_load(reinterpret_cast<const lpx_d2*>(p));
        return make_double2(v.x, v.y);
    } else {
        return *reinterpret_cast<const double2*>(p);
    }
}
template <bool STREAM> __device__ __forceinline__ void upd_store(double* p, double2 o)
{
    if constexpr (STREAM) {
        lpx_d2 v; v.x = o.x; v.y = o.y;
        __builtin_nontemporal_store(v, reinterpret_cast<lpx_d2*>(p));
    } else {
        *reinterpret_cast<double2*>(p) = o;
    }
}

// The same through pointers KNOWN to be global memory.  A kernel that takes its buffers from a parameter record in memory (the
// batched group kernels) sees generic pointers and would issue flat_load / flat_store, which count against both the vector-memory
// and the LDS counter; casting to address space 1 gives global_load_dwordx4 / global_store_dwordx4 as in the single-tableau kernels.
#define LPX_GLOBAL __attribute__((address_space(1)))
template <bool STREAM> __device__ __forceinline__ double2 upd_load_g(const LPX_GLOBAL double* p)
{
    const LPX_GLOBAL lpx_d2* q = (const LPX_GLOBAL lpx_d2*)p;
    lpx_d2 v;
    if constexpr (STREAM) v = __builtin_nontemporal_load(q); else v = *q;
    return make_double2(v.x, v.y);
}
template <bool STREAM> __device__ __forceinline__ void upd_store_g(LPX_GLOBAL double* p, double2 o)
{
    lpx_d2 v; v.x = o.x; v.y = o.y;
    LPX_GLOBAL lpx_d2* q = (LPX_GLOBAL lpx_d2*)p;
    if constexpr (STREAM) __builtin_nontemporal_store(v, q); else *q = v;
}

template <int ROWS = UPD_ROWS, int NTH = UPD_NT, int POLICY = 0>
__device__ __forceinline__ void lpx_update_body(double* __restrict__ T, int ld, int Rcap, int Ccap,
                                                const int32_t* __restrict__ shape,
                                                const double* __restrict__ prow,
                                                double* fac0, double* fac1,
                                                double* __restrict__ rhsbuf,
                                                const DevState* __restrict__ st,
                                                int ncw, int nunits, int mixmod = 1)
{
    if (st->status != LPX_RUNNING) return;
    const int r = st->r;
    if (r < 0) return;
    const int R = shape ? shape[0] : Rcap, C = shape ? shape[1] : Ccap;
    const int par = (st->iter - 1) & 1;                   // parity of the pivot being applied
    const double* __restrict__ fac = par ? fac1 : fac0;   // pivot column snapshot (factors)
    double* __restrict__ nxt = par ? fac0 : fac1;         // by-product: next pivot's column
    const int qn = st->qn;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int unit = blockIdx.x * (NTH / 64) + wave;
    if (unit >= nunits) return;
    const int cw = unit % ncw;
    const int rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;                      // ld is a multiple of 16, so col+1 < ld too
    const double2 p = *reinterpret_cast<const double2*>(prow + col);
    const int row0 = rb * ROWS;
    if (row0 >= R) return;
    double* base = T + (size_t)row0 * ld + col;
    constexpr bool NT = POLICY != 0;
    const int c0 = cw * 128;
    // straight-line path of almost every wave (see lpx_update_mb_body): all rows live, no pivot row, no column to capture
    const bool plain_wave = row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS) &&
                            !(qn >= c0 && qn < c0 + 128) && !(rhsbuf != nullptr && C - 1 >= c0 && C - 1 < c0 + 128);
    if (NT && plain_wave) {
        double2 v[ROWS];
        double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = upd_load<NT>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            v[k].x = v[k].x - f[k] * p.x;       // mul, then sub: contraction is off
            v[k].y = v[k].y - f[k] * p.y;
        }
        // mixed form: the LAST row of the wave goes through the Infinity Cache (default policy) in every `mixmod`-th row
        // block, everything else is non-temporal; two spelled-out sequences so that no store loses its policy when the
        // compiler merges code (checked in the ISA: hipcc keeps `nt` as metadata only)
        if (POLICY == 2 && (mixmod <= 1 || rb % mixmod == 0)) {
#pragma unroll
            for (int k = 0; k < ROWS - 1; ++k) upd_store<true>(base + (size_t)k * ld, v[k]);
            upd_store<false>(base + (size_t)(ROWS - 1) * ld, v[ROWS - 1]);
        } else {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) upd_store<NT>(base + (size_t)k * ld, v[k]);
        }
        return;
    }
    // every other wave -- and every wave of the cache-resident form (8 rows x 256 lanes, default policy), which measured
    // FASTER with the per-row branches (lpx_update_b 53 vs 74 us, config 2's update 9.1 vs 16.4 us): all loads, then per row
    const bool wq = (qn >= 0) && ((qn & ~1) == col);              // this lane owns column qn
    const bool wr = (rhsbuf != nullptr) && (((C - 1) & ~1) == col);   // this lane owns the RHS column
    double2 v[ROWS];
    double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            v[k] = upd_load<NT>(base + (size_t)k * ld);
            f[k] = fac[i];
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) {
            double2 o;
            o.x = v[k].x - f[k] * p.x;          // mul, then sub: contraction is off
            o.y = v[k].y - f[k] * p.y;
            upd_store<NT>(base + (size_t)k * ld, o);
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
            if (wr) rhsbuf[i] = ((C - 1) & 1) ? o.y : o.x;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Multi-workgroup select (primal and forced paths).
//
// lpx_select_la above is one 1024-lane workgroup: ~9.5 us per pivot at 1025x3073, of which the
// barrier-heavy ratio scan is 4 us and the 3073 divisions plus the wait for the slowest wave another
// 4.6 us (in-kernel stamps, tools/diag_select_stamps.py).  Here `nblk` 256-lane workgroups run the
// same step: every workgroup repeats the ratio test (contiguous 8 B x 2 x m, L2-resident, identical
// result everywhere), then normalises ONE column slice of the pivot row, scans the same slice of the
// updated objective row and publishes a partial argmin; the workgroup that arrives last reduces the <= 128
// partials to the next entering column (one word the update kernel reads).
//
// Two state records break what would otherwise be intra-kernel races:
//   st  written by select workgroup 0, read by every update workgroup and by the host;
//   us  written by update workgroup 0, read by every select workgroup.
// Apart from the partials (agent-scope stores and loads around one agent-scope counter, see below) nothing is read
// and written by workgroups of the same launch; the kernel boundary orders the rest (cdna_hip_programming.md G16).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lpx_select_mb_body(const SelParams& P)
{
    __shared__ double s_v[MB_NT / 64];
    __shared__ int s_i[MB_NT / 64];

    const DevState* us = P.us;
    DevState* st = P.st;
    const int t = threadIdx.x, b = blockIdx.x;
    LPX_STAMP_BEGIN
    const int status_in = us->status;
    if (status_in != LPX_RUNNING) { if (b == 0 && t == 0) st->status = status_in; return; }

    LPX_STAMP_MB(0);
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    const int m = R - 1;
    const size_t ld = (size_t)P.ld;
    double* T = P.T;
    const int iter = us->iter;
    const int primal_count = us->primal_count;
    double* colc = (iter & 1) ? P.col1 : P.col0;     // column q of the current tableau
    const int q = us->qn;
    int r = -1, scanrow = -1;
    int final_status = LPX_RUNNING;
    ScanRule rule; rule.forced = (P.mode == MODE_FORCED); rule.eps = P.eps; rule.thresh = P.fthresh;
    rule.C = C; rule.c0 = 0;
    const int k = us->forced_k;

    if (rule.forced) {
        if (k >= P.fcount) {
            final_status = LPX_OPTIMAL;
        } else {
            r = P.frows[k];
            scanrow = (k + 1 < P.fcount) ? P.frows[k + 1] : -1;
            rule.c0 = (k + 1 < P.fcount) ? P.fcols[k + 1] : 0;
            if (q < 0) {                                  // no eligible column: skip this pivot
                if (b != 0) return;
                int qn = la_prepare_from_T<MB_NT>(P, R, C, colc, scanrow, rule, s_v, s_i);
                if (t == 0) {
                    P.fchosen[k] = -1;
                    st->status = LPX_RUNNING; st->iter = iter; st->r = -1; st->q = -1;
                    st->forced_k = k + 1; st->primal_count = primal_count;
                    st->qn = qn; st->qn_valid = 1; st->c0n = rule.c0;
                }
                return;
            }
        }
    } else {
        // primal loop head, Models/PrimalSimplex.cs:95-106
        if (primal_count >= P.max_iter) final_status = LPX_ITER_LIMIT;
        else if (q < 0) final_status = LPX_OPTIMAL;
        else {
            // every wave of every workgroup gets the same row: one wave each for m <= 1024 (no barrier), the segments of a
            // longer column spread over the workgroup's waves
            r = block_hysteresis_segments<MB_NT / 64>(m, P.tol_primal, RowRatio{colc, 1, P.rhsbuf, 1, P.eps});
            if (r < 0) final_status = LPX_UNBOUNDED;
            scanrow = m;
        }
    }
    LPX_STAMP_MB(1);
    if (final_status != LPX_RUNNING) {
        if (b == 0 && t == 0) { st->status = final_status; st->iter = iter; st->r = -1; st->q = -1; }
        return;
    }

    // this workgroup's column slice
    const int per = (C + P.nblk - 1) / P.nblk;
    const int j0 = b * per, j1 = min(C, j0 + per);
    const double piv = colc[r];
    const bool same = (scanrow == r);
    const double fs = (scanrow >= 0 && !same) ? colc[scanrow] : 0.0;
    double* trow = T + (size_t)r * ld;
    const double* srow = T + (size_t)(scanrow >= 0 ? scanrow : 0) * ld;
    MinIdx best; rule_init(rule, best);
    for (int j = j0 + t; j < j1; j += MB_NT) {
        const double p = trow[j] / piv;                       // true division, :250
        trow[j] = p;
        P.prow[j] = p;
        if (scanrow >= 0) {
            const double u = same ? p : srow[j] - fs * p;     // what lpx_update will store at T[s,j]
            rule_feed(rule, best, j, u);
        }
    }
    LPX_STAMP_MB(2);
    best = wave_min_idx(best);                               // one partial per wave
    LPX_STAMP_MB(3);
    if (scanrow >= 0 && !P.qsel) {
        if ((t & 63) == 0) { P.part_v[b * (MB_NT / 64) + (t >> 6)] = best.v; P.part_i[b * (MB_NT / 64) + (t >> 6)] = best.i; }
    } else if (scanrow >= 0) {
        // Streaming tableaux: the workgroup that arrives LAST reduces the partials to the next entering column, so that the update kernel's
        // 10^5 waves read one word instead of each repeating a 128-entry reduction in front of their tile (that prologue
        // cost the streaming update 10 % of its bandwidth, tools/kbench/sweep_dir.hip).  Hand-off as MI355X_MICROARCH.md
        // prescribes for cross-XCD data without fences: every partial is an agent-scope (sc1) store, every storing wave
        // waits for its stores, a workgroup barrier, ONE agent-scope add per workgroup; the workgroup whose add returns
        // nblk - 1 loads the partials with agent-scope (sc1) loads, after its add has returned.
        if ((t & 63) == 0) {
            const int slot = b * (MB_NT / 64) + (t >> 6);
            __hip_atomic_store(&P.part_v[slot], best.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P.part_i[slot], best.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t < 64) {
            int last = 0;
            if (t == 0) last = (__hip_atomic_fetch_add(&P.part_i[MB_CNT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == P.nblk - 1) ? 1 : 0;
            last = __builtin_amdgcn_readfirstlane(last);
            if (last) {
                MinIdx x; x.v = rule.forced ? 0.0 : __builtin_inf(); x.i = INT_MAX;
                const int npart = P.nblk * (MB_NT / 64);               // <= 128
                if (t < npart) {
                    x.v = __hip_atomic_load(&P.part_v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x.i = __hip_atomic_load(&P.part_i[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (t + 64 < npart) {
                    MinIdx y;
                    y.v = __hip_atomic_load(&P.part_v[t + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    y.i = __hip_atomic_load(&P.part_i[t + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    x = mi_pick(x, y);
                }
                x = wave_min_idx(x);
                int qn;
                if (x.i == INT_MAX) qn = -1;
                else if (rule.forced) { qn = rule.c0 + x.i; if (qn >= C) qn -= C; }
                else qn = x.i;
                if (t == 0) {
                    P.part_i[MB_QREC] = qn;                                // read by the update kernel (next launch)
                    __hip_atomic_store(&P.part_i[MB_CNT], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
    if (t == 0) {
        if (b == 0) {
            if (rule.forced) P.fchosen[k] = q; else P.basis[r] = q;     // basis[leaving] = entering, :110
            if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
            st->status = LPX_RUNNING;
            st->iter = iter + 1; st->r = r; st->q = q;
            st->primal_count = rule.forced ? primal_count : primal_count + 1;
            st->forced_k = rule.forced ? k + 1 : k;
            st->qn = -1; st->qn_valid = (scanrow >= 0) ? 0 : 1;         // no scan row: next column is "none"
            st->c0n = rule.c0;
        }
    }
    LPX_STAMP_MB(4);
    LPX_STAMP_END_MB;
}

// lpx_update for the multi-workgroup protocol: same streaming body; the next entering column comes from
// the select workgroups' partials, and workgroup 0 commits the state record `us` for the next select.
// POLICY: 0 = default cache policy (tableau at home in the Infinity Cache), 1 = nontemporal loads and stores, 2 =
// nontemporal loads, the wave's FIRST row stored with the default policy and the others nontemporal (see UPDM below).
template <int ROWS = UPD_ROWS, int NTH = UPD_NT, int POLICY = 0>
__device__ __forceinline__ void lpx_update_mb_body(double* __restrict__ T, int ld, int Rcap, int Ccap,
                                                   const int32_t* __restrict__ shape,
                                                   const double* __restrict__ prow,
                                                   double* fac0, double* fac1,
                                                   double* __restrict__ rhsbuf,
                                                   const DevState* __restrict__ st, DevState* us,
                                                   const double* __restrict__ part_v,
                                                   const int32_t* __restrict__ part_i, int nblk,
                                                   int forced, int ncw, int nunits, int mixmod = 1)
{
    constexpr bool NT = POLICY != 0;
    // The state record is read FIRST: a launch that finds the loop finished (tail of a batch, finished node
    // of a B&B group) must not stream the tableau.
    const int status = st->status;
    const int r = st->r;
    const int lane = threadIdx.x & 63;
    const int R = shape ? shape[0] : Rcap, C = shape ? shape[1] : Ccap;
    if (status != LPX_RUNNING) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { us->status = status; us->iter = st->iter; }
        return;
    }
    // next entering column: set by select directly (skipped pivot / no scan row), or the minimum of the select workgroups'
    // partials.  Streaming forms: the select workgroup that finished last has reduced them to one word (SelParams::qsel)
    // -- 10^5 single-wave workgroups repeating the reduction in front of their tile cost the update 10 % of its bandwidth.
    // Cache-resident form: every wave reduces them here (one load + 6 DPP steps), which is cheaper than the 1.5-2 us the
    // last-workgroup hand-off adds to select when a pivot takes 15 us in all.
    int qn;
    if (st->qn_valid) {
        qn = st->qn;
    } else if constexpr (POLICY != 0) {
        qn = part_i[MB_QREC];
    } else {
        MinIdx x; x.v = forced ? 0.0 : __builtin_inf(); x.i = INT_MAX;
        const int npart = nblk * (MB_NT / 64);               // <= 128: one partial per select wave
        if (lane < npart) { x.v = part_v[lane]; x.i = part_i[lane]; }
        if (lane + 64 < npart) { MinIdx y; y.v = part_v[lane + 64]; y.i = part_i[lane + 64]; x = mi_pick(x, y); }
        x = wave_min_idx(x);
        if (x.i == INT_MAX) qn = -1;
        else if (forced) { qn = st->c0n + x.i; if (qn >= C) qn -= C; }
        else qn = x.i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        us->status = LPX_RUNNING; us->iter = st->iter; us->qn = qn;
        us->primal_count = st->primal_count; us->forced_k = st->forced_k;
    }
    if (r < 0) return;                                   // skipped pivot: nothing to update
    const int par = (st->iter - 1) & 1;
    const double* __restrict__ fac = par ? fac1 : fac0;
    double* __restrict__ nxt = par ? fac0 : fac1;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int unit = blockIdx.x * (NTH / 64) + wave;
    if (unit >= nunits) return;
    const int cw = unit % ncw;
    const int rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    if (row0 >= R) return;                                // capacity-sized grid: rows beyond the live shape
    const double2 p = *reinterpret_cast<const double2*>(prow + col);
    double* base = T + (size_t)row0 * ld + col;
    const int c0 = cw * 128;
    // Almost every wave takes the straight-line path: all ROWS rows live, none of them the pivot row, no column to
    // capture.  Loads, arithmetic and stores follow each other without a branch, so the wait counts stay exact (with
    // a branch per row the compiler waits for EVERYTHING, the previous row's store acknowledgement included, before
    // each store).
    const bool plain_wave = row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS) &&
                            !(qn >= c0 && qn < c0 + 128) && !(C - 1 >= c0 && C - 1 < c0 + 128);
    if (NT && plain_wave) {
        double2 v[ROWS];
        double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = upd_load<NT>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            v[k].x = v[k].x - f[k] * p.x;       // mul, then sub: contraction is off
            v[k].y = v[k].y - f[k] * p.y;
        }
        // mixed form: the LAST row of the wave goes through the Infinity Cache (default policy) in every `mixmod`-th row
        // block, everything else is non-temporal; two spelled-out sequences so that no store loses its policy when the
        // compiler merges code (checked in the ISA: hipcc keeps `nt` as metadata only)
        if (POLICY == 2 && (mixmod <= 1 || rb % mixmod == 0)) {
#pragma unroll
            for (int k = 0; k < ROWS - 1; ++k) upd_store<true>(base + (size_t)k * ld, v[k]);
            upd_store<false>(base + (size_t)(ROWS - 1) * ld, v[ROWS - 1]);
        } else {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) upd_store<NT>(base + (size_t)k * ld, v[k]);
        }
        return;
    }
    // every other wave, and every wave of the cache-resident form (which measured faster with the per-row branches)
    const bool wq = (qn >= 0) && ((qn & ~1) == col);
    const bool wr = (((C - 1) & ~1) == col);
    double2 v[ROWS];
    double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            v[k] = upd_load<NT>(base + (size_t)k * ld);
            f[k] = fac[i];
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            double2 o;
            if (i != r) {
                o.x = v[k].x - f[k] * p.x;      // mul, then sub: contraction is off
                o.y = v[k].y - f[k] * p.y;
                upd_store<NT>(base + (size_t)k * ld, o);
            } else {
                o = p;                          // row r already holds the normalised pivot row
            }
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
            if (wr) rhsbuf[i] = ((C - 1) & 1) ? o.y : o.x;
        }
    }
}

// single-tableau and batched (blockIdx.y = node of a branch-and-bound group) entry points
__global__ __launch_bounds__(SEL_NT) void lpx_select(SelParams P, int lds_doubles)
{
    extern __shared__ __align__(16) double sel_lds[];
    lpx_select_body(P, lds_doubles > 0 ? sel_lds : nullptr);
}
__global__ __launch_bounds__(SEL_NT) void lpx_rhs_init(SelParams P) { lpx_rhs_init_body(P); }
__global__ __launch_bounds__(SEL_NT) void lpx_rhs_init_b(const SelParams* __restrict__ arr) { const SelParams P = arr[blockIdx.y]; lpx_rhs_init_body(P); }
__global__ __launch_bounds__(SEL_NT) void lpx_la_init(SelParams P) { lpx_la_init_body(P); }
__global__ __launch_bounds__(MB_NT) void lpx_select_mb(SelParams P) { lpx_select_mb_body(P); }
__global__ __launch_bounds__(SEL_NT) void lpx_select_b(const SelParams* __restrict__ arr, int lds_doubles)
{
    extern __shared__ __align__(16) double sel_lds[];
    const SelParams P = arr[blockIdx.y];
    lpx_select_body(P, lds_doubles > 0 ? sel_lds : nullptr);
}
__global__ __launch_bounds__(SEL_NT) void lpx_la_init_b(const SelParams* __restrict__ arr) { const SelParams P = arr[blockIdx.y]; lpx_la_init_body(P); }
__global__ __launch_bounds__(MB_NT) void lpx_select_mb_b(const SelParams* __restrict__ arr)
{
    const SelParams P = arr[blockIdx.y];
    if ((int)blockIdx.x >= P.nblk) return;          // grid is sized for the widest node of the group
    lpx_select_mb_body(P);
}

__global__ __launch_bounds__(UPD_NT) void lpx_update(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                     const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                     const DevState* st, int ncw, int nunits)
{
    lpx_update_body(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits);
}
__global__ __launch_bounds__(UPD_NT) void lpx_update_mb(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                        const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                        const DevState* st, DevState* us, const double* part_v,
                                                        const int32_t* part_i, int nblk, int forced, int ncw, int nunits)
{
    lpx_update_mb_body(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, us, part_v, part_i, nblk, forced, ncw, nunits);
}
// streaming variants (tableau larger than the Infinity Cache): see UPDS_ROWS above
__global__ __launch_bounds__(UPDS_NT) void lpx_update_s(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                        const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                        const DevState* st, int ncw, int nunits, int mixmod)
{
    lpx_update_body<UPDS_ROWS, UPDS_NT, 1>(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits, mixmod);
}
__global__ __launch_bounds__(UPDS_NT) void lpx_update_m(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                        const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                        const DevState* st, int ncw, int nunits, int mixmod)
{
    lpx_update_body<UPDS_ROWS, UPDS_NT, 2>(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits, mixmod);
}
__global__ __launch_bounds__(UPDS_NT) void lpx_update_mb_s(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                           const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                           const DevState* st, DevState* us, const double* part_v,
                                                           const int32_t* part_i, int nblk, int forced, int ncw, int nunits, int mixmod)
{
    lpx_update_mb_body<UPDS_ROWS, UPDS_NT, 1>(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, us, part_v, part_i, nblk, forced, ncw, nunits, mixmod);
}
__global__ __launch_bounds__(UPDS_NT) void lpx_update_mb_m(double* T, int ld, int Rcap, int Ccap, const int32_t* shape,
                                                           const double* prow, double* fac0, double* fac1, double* rhsbuf,
                                                           const DevState* st, DevState* us, const double* part_v,
                                                           const int32_t* part_i, int nblk, int forced, int ncw, int nunits, int mixmod)
{
    lpx_update_mb_body<UPDS_ROWS, UPDS_NT, 2>(T, ld, Rcap, Ccap, shape, prow, fac0, fac1, rhsbuf, st, us, part_v, part_i, nblk, forced, ncw, nunits, mixmod);
}
// batched: every node of the group advances by one pivot per launch pair; grid.x covers the largest node
__global__ __launch_bounds__(UPD_NT) void lpx_update_b(const SelParams* __restrict__ arr)
{
    const SelParams P = arr[blockIdx.y];
    const int ncw = (P.ld + 127) / 128, nunits = ncw * ((P.R + UPD_ROWS - 1) / UPD_ROWS);
    lpx_update_body(P.T, P.ld, P.R, P.C, P.shape, P.prow, P.pcol, P.pcol, P.rhsbuf, P.st, ncw, nunits);
}
__global__ __launch_bounds__(UPD_NT) void lpx_update_mb_b(const SelParams* __restrict__ arr)
{
    const SelParams P = arr[blockIdx.y];
    const int ncw = (P.ld + 127) / 128, nunits = ncw * ((P.R + UPD_ROWS - 1) / UPD_ROWS);
    lpx_update_mb_body(P.T, P.ld, P.R, P.C, P.shape, P.prow, P.col0, P.col1, P.rhsbuf, P.st, P.us, P.part_v, P.part_i,
                       P.nblk, P.mode == MODE_FORCED ? 1 : 0, ncw, nunits);
}

// ------------------------------------------------------------------------------------------------
// Branch-and-bound node assembly on the device.  A node LP is the root model plus `d` unit rows
// (Models/Branch&Bound.cs:233-248); its tableau (BuildTableau, Models/PrimalSimplex.cs:179-203) is the
// root tableau with d more rows and d more slack columns.  The root tableau stays resident; a node is
// built by one streaming kernel from it and d cut descriptors instead of 8 MB of host work + H2D.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lpx_build_node(const double* __restrict__ T0, int ld0, int R0, int C0,
                                                      double* __restrict__ T, int ld, int R, int C,
                                                      const int32_t* __restrict__ cvar, const double* __restrict__ ccoef,
                                                      const double* __restrict__ czero, const double* __restrict__ crhs,
                                                      int32_t* __restrict__ basis)
{
    const int m0 = R0 - 1, m = R - 1, n = C0 - R0, d = R - R0;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j == 0 && i < m) basis[i] = n + i;                               // :197
    if (j >= ld) return;
    double v = 0.0;
    if (j < C) {
        if (i < m0 || i == m) {                                          // root constraint rows / objective row
            const int i0 = (i == m) ? m0 : i;
            if (j < n + m0) v = T0[(size_t)i0 * ld0 + j];
            else if (j == C - 1) v = T0[(size_t)i0 * ld0 + (C0 - 1)];
        } else {                                                         // branching row k
            const int k = i - m0;
            if (j < n) v = (j == cvar[k]) ? ccoef[k] : czero[k];
            else if (j == n + m0 + k) v = 1.0;                           // its slack, :191
            else if (j == C - 1) v = crhs[k];                            // :192
        }
    }
    (void)d;
    T[(size_t)i * ld + j] = v;
}

// The same for a whole group of nodes in ONE launch (blockIdx.z = node): small node LPs are solved hundreds at a time, and
// a launch + two small copies per node were the largest host phase left.  The kernel also writes each node's live-shape
// record and clears its state record.
__global__ __launch_bounds__(256) void lpx_build_nodes(const double* __restrict__ T0, int ld0, int R0, int C0,
                                                       const BuildDesc* __restrict__ descs,
                                                       const int32_t* __restrict__ cvar, const double* __restrict__ ccoef,
                                                       const double* __restrict__ czero, const double* __restrict__ crhs)
{
    const BuildDesc D = descs[blockIdx.z];
    const int R = D.R, C = D.C, ld = D.ld;
    const int m0 = R0 - 1, m = R - 1, n = C0 - R0;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= R) return;
    if (i == 0 && blockIdx.x == 0) {
        if (threadIdx.x == 0) { D.shape[0] = R; D.shape[1] = C; }
        int32_t* stw = reinterpret_cast<int32_t*>(D.st);
        for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 256) stw[k] = 0;
    }
    if (j == 0 && i < m) D.basis[i] = n + i;                             // :197
    if (j >= ld) return;
    double v = 0.0;
    if (j < C) {
        if (i < m0 || i == m) {                                          // root constraint rows / objective row
            const int i0 = (i == m) ? m0 : i;
            if (j < n + m0) v = T0[(size_t)i0 * ld0 + j];
            else if (j == C - 1) v = T0[(size_t)i0 * ld0 + (C0 - 1)];
        } else {                                                         // branching row k
            const int k = D.cut0 + (i - m0);
            if (j < n) v = (j == cvar[k]) ? ccoef[k] : czero[k];
            else if (j == n + m0 + (i - m0)) v = 1.0;                    // its slack, :191
            else if (j == C - 1) v = crhs[k];                            // :192
        }
    }
    D.T[(size_t)i * ld + j] = v;
}

hipError_t launch_build_nodes(const double* T0, int ld0, int R0, int C0, const BuildDesc* descs, int count, int maxld, int maxR,
                              const int32_t* cvar, const double* ccoef, const double* czero, const double* crhs, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_build_nodes, dim3((maxld + 255) / 256, maxR, count), dim3(256), 0, s, T0, ld0, R0, C0, descs, cvar, ccoef, czero, crhs);
    return hipGetLastError();
}

// Warm start: the child of a solved node is the parent's FINAL tableau plus one branching
// row expressed in the parent's basis.  With x_k basic in row ik:  `x_k <= f`  becomes  e_k - T[ik,:]  (rhs
// f - x_k* < 0) and  `x_k >= c`  becomes  -e_k + T[ik,:]  (rhs -c + x_k* < 0); the new slack is basic in the new
// row.  The objective row is unchanged, so the tableau stays dual feasible and only the dual loop has to run.
__global__ __launch_bounds__(256) void lpx_build_child(const double* __restrict__ Tp, int ldp, int Rp, int Cp,
                                                       const int32_t* __restrict__ basis_p,
                                                       double* __restrict__ T, int ld, int var, int ik, int is_ge,
                                                       double bound, int32_t* __restrict__ basis)
{
    const int mp = Rp - 1, R = Rp + 1, C = Cp + 1;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (j == 0 && i < mp) basis[i] = basis_p[i];
    if (j == 0 && i == mp) basis[mp] = Cp - 1;                      // the new slack column
    if (j >= ld || i >= R) return;
    double v = 0.0;
    if (j < C) {
        const int js = (j < Cp - 1) ? j : ((j == C - 1) ? Cp - 1 : -1);   // source column in the parent (-1: new slack)
        if (i == mp) {                                              // the branching row
            if (js < 0) v = 1.0;
            else {
                const double t = Tp[(size_t)ik * ldp + js];
                const double e = (js == var) ? 1.0 : 0.0;
                const double rhs = (js == Cp - 1) ? bound : 0.0;
                v = is_ge ? ((-e - rhs) + t) : ((e + rhs) - t);     // GE: -e_k + row, rhs -c + x_k ; LE: e_k - row, rhs f - x_k
            }
        } else {
            const int is = (i < mp) ? i : mp;                       // i == mp + 1 is the parent's objective row
            v = (js < 0) ? 0.0 : Tp[(size_t)is * ldp + js];
        }
    }
    T[(size_t)i * ld + j] = v;
}

// lpx_build_child for a group of children in one launch (blockIdx.z = child); shape and state records written here too
__global__ __launch_bounds__(256) void lpx_build_children(const ChildDesc* __restrict__ descs)
{
    const ChildDesc D = descs[blockIdx.z];
    const double* __restrict__ Tp = D.Tp; double* __restrict__ T = D.T;
    const int ldp = D.ldp, Rp = D.Rp, Cp = D.Cp, ld = D.ld, var = D.var, ik = D.ik, is_ge = D.is_ge;
    const double bound = D.bound;
    const int mp = Rp - 1, R = Rp + 1, C = Cp + 1;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= R) return;
    if (i == 0 && blockIdx.x == 0) {
        if (threadIdx.x == 0) { D.shape[0] = R; D.shape[1] = C; }
        int32_t* stw = reinterpret_cast<int32_t*>(D.st);
        for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 256) stw[k] = 0;
    }
    if (j == 0 && i < mp) D.basis[i] = D.basis_p[i];
    if (j == 0 && i == mp) D.basis[mp] = Cp - 1;                    // the new slack column
    if (j >= ld) return;
    double v = 0.0;
    if (j < C) {
        const int js = (j < Cp - 1) ? j : ((j == C - 1) ? Cp - 1 : -1);   // source column in the parent (-1: new slack)
        if (i == mp) {                                              // the branching row
            if (js < 0) v = 1.0;
            else {
                const double t = Tp[(size_t)ik * ldp + js];
                const double e = (js == var) ? 1.0 : 0.0;
                const double rhs = (js == Cp - 1) ? bound : 0.0;
                v = is_ge ? ((-e - rhs) + t) : ((e + rhs) - t);     // GE: -e_k + row, rhs -c + x_k ; LE: e_k - row, rhs f - x_k
            }
        } else {
            const int is = (i < mp) ? i : mp;                       // i == mp + 1 is the parent's objective row
            v = (js < 0) ? 0.0 : Tp[(size_t)is * ldp + js];
        }
    }
    T[(size_t)i * ld + j] = v;
}

hipError_t launch_build_children(const ChildDesc* descs, int count, int maxld, int maxR, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_build_children, dim3((maxld + 255) / 256, maxR, count), dim3(256), 0, s, descs);
    return hipGetLastError();
}

hipError_t launch_build_child(const double* Tp, int ldp, int Rp, int Cp, const int32_t* basis_p, double* T, int ld,
                              int var, int ik, int is_ge, double bound, int32_t* basis, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_build_child, dim3((ld + 255) / 256, Rp + 1), dim3(256), 0, s, Tp, ldp, Rp, Cp, basis_p, T, ld,
                       var, ik, is_ge, bound, basis);
    return hipGetLastError();
}

// Final solution of a whole batch of nodes in one launch (FinalizeReport's reads, Models/PrimalSimplex.cs:135-138, for
// every node of a B&B group): block b copies node b's RHS column and basis into one contiguous record of the output,
// which is pinned host memory the kernel writes directly -- one launch + one wait per batch instead of two strided
// copies + one wait per node.
__global__ __launch_bounds__(256) void lpx_gather_solution(const GatherDesc* __restrict__ descs, double* __restrict__ out_rhs,
                                                           int32_t* __restrict__ out_basis)
{
    const GatherDesc D = descs[blockIdx.x];
    for (int i = threadIdx.x; i < D.R; i += 256) {
        out_rhs[D.off + i] = D.T[(size_t)i * D.ld + (D.C - 1)];
        if (i < D.R - 1) out_basis[D.off + i] = D.basis[i];
    }
}

hipError_t launch_gather_solution(const GatherDesc* descs, int count, double* out_rhs, int32_t* out_basis, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_gather_solution, dim3(count), dim3(256), 0, s, descs, out_rhs, out_basis);
    return hipGetLastError();
}

hipError_t launch_build_node(const double* T0, int ld0, int R0, int C0, double* T, int ld, int R, int C,
                             const int32_t* cvar, const double* ccoef, const double* czero, const double* crhs,
                             int32_t* basis, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_build_node, dim3((ld + 255) / 256, R), dim3(256), 0, s, T0, ld0, R0, C0, T, ld, R, C,
                       cvar, ccoef, czero, crhs, basis);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fused pivot (primal loop without a per-pivot callback, any size): update(k) OUT OF PLACE + select(k+1), one launch.
//
// With the update in place, select(k+1) has to wait for update(k): it reads column q', row r' and the objective row of
// T_{k+1}.  All three are rank-1 corrections of the same parts of T_k with data select(k) already produced (the factor
// column and the normalised pivot row of pivot k) -- so if update(k) writes T_{k+1} into a SECOND buffer and leaves T_k
// alone, select(k+1) depends on nothing update(k) produces and runs beside it: the first `nblk` workgroups of this grid are
// lpx_select_mb's workgroups reading T_k through the correction `T_k[i,j] - fac_k[i] * prow_k[j]` (the very mul-then-sub
// the update stores, so every value is bit-identical to what the two-launch path reads back from memory), the others are
// the streaming update's waves.  A pivot then costs the sweep alone (117.7 us at 4097 x 12289 against 117.4 + 10.0);
// tools/kbench/oop.hip measured the shape first: a ping-pong sweep is as fast as the in-place one (115.5 vs 116.0 us),
// 256-lane workgroups cost it 1.6 us, and 32 select-shaped workgroups at the head of the grid another 0.5 us.
// Smaller tableaux gain more (tools/probe_fused_mid.py: 1.03-1.42x from 129 x 385 to 308 MB); cache policy: fused_policy below.
//
// Nothing is read and written inside one launch: everything a launch reads carries the index `c` of the CURRENT state
// record (pivot row, factor column, RHS column) or is the source tableau; everything it writes carries 1 - c or is the
// destination tableau -- the state records included: a launch reads record `par` and writes record 1 - par, and `par` is a
// launch argument that alternates (graph batches are even, so a replay starts where the capture did).  Which buffer holds
// T_k is part of the record (pad[3]); pad[2] counts launches, for the host to find the last record written.
//   record c:  (r, q) = pivot k, selected but NOT yet applied ("pending"; r < 0: none)   qn = entering column of pivot k+1
//   prow[c] = T_k[r,:] / T_k[r,q]     col[c] = T_k[:,q]     rhs[c] = T_k[:,C-1]
// P.st is the host's copy of the current record, written by the first update workgroup: one launch behind.
// A terminal status is found by select(k+1) in the launch that applies pivot k, so the tableau is complete when it shows.
// ------------------------------------------------------------------------------------------------
static constexpr int FP_NT = 256;

// the ratios of pivot k+1's test, formed once per workgroup into its slice of P.ws: the scan then holds 16 ratios per lane
// and nothing else (a scan over T_k's strided column with the correction applied on the fly needed 168 VGPRs, which
// left the update waves of the same kernel 3 waves per SIMD and the sweep 10 us slower)
struct CompactRatio {
    const double* rat;
    __device__ __forceinline__ double den(int i) const { return rat[i]; }
    __device__ __forceinline__ double num(int) const { return 0.0; }
    __device__ __forceinline__ double value(double a, double) const { return a; }
};

__global__ __launch_bounds__(SEL_NT) void lpx_fused_init(FusedParams F)
{
    __shared__ double s_v[SEL_NW];
    __shared__ int s_i[SEL_NW];
    const SelParams& P = F.P;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    ScanRule rule; rule.forced = 0; rule.eps = P.eps; rule.thresh = P.fthresh; rule.C = C; rule.c0 = 0;
    // first entering column (ChooseEntering on the objective row as it stands) and the RHS column of record 0
    int qn = -1;
    if (P.st->status == LPX_RUNNING) qn = la_prepare_from_T(P, R, C, P.col0, R - 1, rule, s_v, s_i);
    if (threadIdx.x == 0) {
        DevState x = *P.st;
        x.qn = qn; x.r = -1; x.q = -1; x.pad[2] = 1; x.pad[3] = 0;
        F.rec[0] = x;
        x.pad[2] = 0;
        F.rec[1] = x;
        P.part_i[MB_CNT] = 0;
    }
}

// waves_per_eu(6): 78 VGPRs, no spills (86 without the hint = 5 waves per SIMD: 8.46 k pivots/s against 8.57 k; 64 VGPRs with 7 spills: 8.55 k)
// NT: nontemporal loads and (mixmod permitting) stores -- the streaming forms; false: default policy throughout, for a pair of
// buffers that lives in the Infinity Cache together (lpx_pivot_fused_c)
template <bool NT>
__device__ __forceinline__ void lpx_pivot_fused_body(const FusedParams& F, int ncw, int nunits, int mixmod)
{
    const SelParams& P = F.P;
    const int t = threadIdx.x;
    // Which record is current comes with the LAUNCH (F.par, alternating; run_fused), not from the records: the workgroups of a
    // launch start over its whole duration, and one that started after workgroup 0 had written the next record must not
    // take that for the current one.  (A first form compared sequence numbers on the device; it passed every test because
    // select finishes late and the scalar cache kept serving the old line -- and broke when a second process shared the GPU.)
    const int c = F.par & 1;
    const DevState curv = F.rec[c];
    const DevState* cur = &curv;
    DevState* nxt = F.rec + (c ^ 1);
    const int status = cur->status;
    const int pr = cur->r;                                   // pending pivot row (-1: nothing to apply)
    const int seq = cur->pad[2], buf = cur->pad[3];
    const int nsel = P.nblk;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    const size_t ld = (size_t)P.ld;
    const double* __restrict__ src = buf ? F.T1 : P.T;
    double* __restrict__ dst = buf ? P.T : F.T1;
    const double* __restrict__ prowc = c ? F.prow1 : P.prow;
    const double* __restrict__ facc = c ? P.col1 : P.col0;

    if ((int)blockIdx.x >= nsel) {
        // ---------------- update(k): T_{k+1} = T_k - fac (x) prow, row r replaced by the normalised pivot row ----------------
        if ((int)blockIdx.x == nsel && t == 0) *P.st = *cur;
        if (status != LPX_RUNNING || pr < 0) return;
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const int unit = ((int)blockIdx.x - nsel) * (FP_NT / 64) + wave;
        if (unit >= nunits) return;
        const int cw = unit % ncw, rb = unit / ncw;
        const int col = cw * 128 + lane * 2;
        if (col >= P.ld) return;
        const int row0 = rb * UPDS_ROWS;
        if (row0 >= R) return;
        const double2 p = *reinterpret_cast<const double2*>(prowc + col);
        const double* sb = src + (size_t)row0 * ld + col;
        double* db = dst + (size_t)row0 * ld + col;
        if (row0 + UPDS_ROWS <= R && (pr < row0 || pr >= row0 + UPDS_ROWS)) {
            double2 v[UPDS_ROWS];
            double f[UPDS_ROWS];
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) v[k] = upd_load<NT>(sb + (size_t)k * ld);
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) f[k] = facc[row0 + k];
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) {
                v[k].x = v[k].x - f[k] * p.x;       // mul, then sub: contraction is off
                v[k].y = v[k].y - f[k] * p.y;
            }
            // store policies as in lpx_update_mb_m / _s (two spelled-out sequences, checked in the ISA)
            if (NT && mixmod > 0 && (mixmod == 1 || rb % mixmod == 0)) {
#pragma unroll
                for (int k = 0; k < UPDS_ROWS - 1; ++k) upd_store<true>(db + (size_t)k * ld, v[k]);
                upd_store<false>(db + (size_t)(UPDS_ROWS - 1) * ld, v[UPDS_ROWS - 1]);
            } else {
#pragma unroll
                for (int k = 0; k < UPDS_ROWS; ++k) upd_store<NT>(db + (size_t)k * ld, v[k]);
            }
            return;
        }
#pragma unroll 1
        for (int k = 0; k < UPDS_ROWS; ++k) {
            const int i = row0 + k;
            if (i >= R) break;
            double2 o = p;                                   // row r: the normalised pivot row
            if (i != pr) {
                const double2 v = upd_load<NT>(sb + (size_t)k * ld);
                const double f = facc[i];
                o.x = v.x - f * p.x;
                o.y = v.y - f * p.y;
            }
            upd_store<NT>(db + (size_t)k * ld, o);
        }
        return;
    }

    // ---------------- select(k+1) on T_{k+1}, read as T_k with pivot k's correction ----------------
    const int b = blockIdx.x;
    const int nbuf = pr >= 0 ? (buf ^ 1) : buf;              // where T_{k+1} lives once this launch is over
    if (status != LPX_RUNNING) {
        if (b == 0 && t == 0) { DevState x = *cur; x.pad[2] = seq + 1; *nxt = x; }
        return;
    }
    double* __restrict__ prown = c ? P.prow : F.prow1;
    double* __restrict__ facn = c ? P.col0 : P.col1;
    const double* __restrict__ rhsc = c ? F.rhs1 : P.rhsbuf;
    double* __restrict__ rhsn = c ? P.rhsbuf : F.rhs1;
    const int m = R - 1;
    const int iter = cur->iter, primal_count = cur->primal_count;
    const int q = cur->qn;
    int final_status = LPX_RUNNING, r = -1;
    // loop head, Models/PrimalSimplex.cs:95-106
    if (primal_count >= P.max_iter) final_status = LPX_ITER_LIMIT;
    else if (q < 0) final_status = LPX_OPTIMAL;
    double pq = 0.0, prhs = 0.0, fs = 0.0;
    if (final_status == LPX_RUNNING) {
        if (pr >= 0) { pq = prowc[q]; prhs = prowc[C - 1]; }
        // T_{k+1}[i,q] and T_{k+1}[i,C-1] as the update stores them (mul, then sub; row r: the normalised pivot row)
        auto den_of = [&](int i, double v, double f) { const double u = v - f * pq; return pr < 0 ? v : (i == pr ? pq : u); };
        auto num_of = [&](int i, double h, double f) { const double u = h - f * prhs; return pr < 0 ? h : (i == pr ? prhs : u); };
        double* rat = P.ws + (size_t)b * (size_t)(P.R > P.C ? P.R : P.C);
        constexpr int U = 6;                                 // rows in flight per lane: three batches of loads at m = 4096
        for (int i0 = 0; i0 < R; i0 += U * FP_NT) {
            double v[U], f[U], h[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(R - 1, i0 + u * FP_NT + t);    // clamped, not guarded: a guarded load waits for its own branch
                v[u] = src[(size_t)i * ld + q]; f[u] = facc[i]; h[u] = rhsc[i];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * FP_NT + t;
                if (i < R) {
                    const double dn = den_of(i, v[u], f[u]), nm = num_of(i, h[u], f[u]);
                    rat[i] = dn > P.eps ? nm / dn : __builtin_inf();     // ChooseLeaving's ratio, :229-241
                    if (b == 0) { facn[i] = dn; rhsn[i] = nm; }          // factors of pivot k+1, numerators of the test after it
                    if (i == m) fs = dn;                                 // T_{k+1}[m,q]
                }
            }
        }
        // every lane needs fs: row m belongs to exactly one lane of the last batch
        __shared__ double s_fs;
        if (((m % (U * FP_NT)) % FP_NT) == t) s_fs = fs;
        __syncthreads();                                     // the slice of ratios is complete (and visible: same CU)
        fs = s_fs;
        r = block_hysteresis_segments<FP_NT / 64>(m, P.tol_primal, CompactRatio{rat});
        if (r < 0) final_status = LPX_UNBOUNDED;
    }
    if (final_status != LPX_RUNNING) {
        if (b == 0 && t == 0) {
            DevState x = *cur;
            x.status = final_status; x.r = -1; x.q = -1; x.qn = -1; x.pad[2] = seq + 1; x.pad[3] = nbuf;
            *nxt = x;
        }
        return;
    }

    ScanRule rule; rule.forced = 0; rule.eps = P.eps; rule.thresh = P.fthresh; rule.C = C; rule.c0 = 0;
    const int per = (C + nsel - 1) / nsel;
    const int j0 = b * per, j1 = min(C, j0 + per);
    const double fr = facc[r], fm = facc[m];
    const double* trow = src + (size_t)r * ld;
    const double* orow = src + (size_t)m * ld;
    const double pv = trow[q];
    const double piv = pr < 0 ? pv : (r == pr ? pq : pv - fr * pq);      // T_{k+1}[r,q]
    MinIdx best; rule_init(rule, best);
#pragma unroll 2
    for (int j = j0 + t; j < j1; j += FP_NT) {
        const double pc = prowc[j], tv = trow[j], ov = orow[j];
        const double tu = tv - fr * pc, ou = ov - fm * pc;
        const double tr = pr < 0 ? tv : (r == pr ? pc : tu);     // T_{k+1}[r,j]
        const double p = tr / piv;                               // true division, :250
        prown[j] = p;
        const double u = (pr < 0 ? ov : ou) - fs * p;            // what update(k+1) will store at T[m,j]
        rule_feed(rule, best, j, u);
    }
    best = wave_min_idx(best);
    // last-workgroup reduction of the partial argmins: the hand-off of lpx_select_mb (agent-scope stores, wait, barrier, one add)
    if ((t & 63) == 0) {
        const int slot = b * (FP_NT / 64) + (t >> 6);
        __hip_atomic_store(&P.part_v[slot], best.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&P.part_i[slot], best.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t < 64) {
        int last = 0;
        if (t == 0) last = (__hip_atomic_fetch_add(&P.part_i[MB_CNT], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsel - 1) ? 1 : 0;
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            MinIdx x; x.v = __builtin_inf(); x.i = INT_MAX;
            const int npart = nsel * (FP_NT / 64);               // <= 128
            if (t < npart) {
                x.v = __hip_atomic_load(&P.part_v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x.i = __hip_atomic_load(&P.part_i[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (t + 64 < npart) {
                MinIdx y;
                y.v = __hip_atomic_load(&P.part_v[t + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                y.i = __hip_atomic_load(&P.part_i[t + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                x = mi_pick(x, y);
            }
            x = wave_min_idx(x);
            if (t == 0) {
                nxt->qn = (x.i == INT_MAX) ? -1 : x.i;           // the one field of the record this workgroup writes
                __hip_atomic_store(&P.part_i[MB_CNT], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (b == 0 && t == 0) {
        P.basis[r] = q;                                          // basis[leaving] = entering, :110
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        nxt->status = LPX_RUNNING; nxt->iter = iter + 1; nxt->r = r; nxt->q = q;
        nxt->phase = cur->phase; nxt->fdf_count = cur->fdf_count; nxt->dual_iter = cur->dual_iter;
        nxt->primal_count = primal_count + 1; nxt->forced_k = cur->forced_k; nxt->c0n = 0; nxt->qn_valid = 0;
        nxt->pad[0] = cur->pad[0]; nxt->pad[1] = cur->pad[1]; nxt->pad[2] = seq + 1; nxt->pad[3] = nbuf;
    }
}

__global__ __launch_bounds__(FP_NT) __attribute__((amdgpu_waves_per_eu(6))) void lpx_pivot_fused(FusedParams F, int ncw, int nunits, int mixmod)
{ lpx_pivot_fused_body<true>(F, ncw, nunits, mixmod); }
__global__ __launch_bounds__(FP_NT) __attribute__((amdgpu_waves_per_eu(6))) void lpx_pivot_fused_c(FusedParams F, int ncw, int nunits, int mixmod)
{ lpx_pivot_fused_body<false>(F, ncw, nunits, mixmod); }

// ------------------------------------------------------------------------------------------------
// Fused GROUP step (K4g): the dual path's three-phase state machine (ForceDualFeasibility Models/DualSimplex.cs:195-228, dual
// loop :36-113, repaired-mode primal clean-up; a primal node is "phase 2 from the start", Models/PrimalSimplex.cs:92-124) for a
// whole group of node LPs in ONE launch per step: update(k) of every live node OUT OF PLACE beside select(k+1) of every live
// node.  The dependency argument is K4f's: select(k+1) reads the RHS column, the objective row, one row and one column of
// T_{k+1}, and each of them is a rank-1 correction of the same part of T_k by data select(k) left behind (factor column, normalised
// pivot row) -- read as `T_k[i,j] - fac[i] * prow[j]`, the very mul-then-sub the update stores, so every value equals bit for bit
// what the two-launch kernels (lpx_select_b + lpx_update_b) read back from memory.
//
// Grid (1-D): the first `nlive` workgroups are the selects, one per live node -- at the head of the grid so that their chain of
// dependent loads runs beside the sweep instead of behind it -- then `per_node` update workgroups per live node (four waves of
// 3 rows x 128 columns each, the streaming tile of lpx_pivot_fused).  `live` maps a slot of the grid to a node of `arr`: finished
// nodes drop out of the grid between polls (the host rewrites the list), they do not cost early-exit workgroups.
// No read-after-write inside a launch: a node's launch reads its record / pivot row / factor column / RHS column of index
// c = (lpar ^ F.par) & 1 and its source tableau, and writes those of index 1 - c and the destination tableau; `lpar` is a launch
// argument that alternates, F.par the node's own offset (set before the run by the host: a node joins a rolling batch at any
// parity).  Scratch (P.ws) belongs to the node's select workgroup alone.
// ------------------------------------------------------------------------------------------------
static constexpr int FG_NT = 256;

__global__ __launch_bounds__(FG_NT) void lpx_group_fused_init(const FusedParams* __restrict__ arr, const int* __restrict__ fresh,
                                                              const DevState* __restrict__ init)
{
    // a node that starts (or continues from another path) in this run: both records from the host's initial state, the RHS
    // column as it stands in the tableau; the node's current record is index F.par (launch 0 of the run has lpar = 0)
    const int k = fresh[blockIdx.x];
    const FusedParams F = arr[k];
    const SelParams& P = F.P;
    const int R = P.shape ? P.shape[0] : P.R, C = P.shape ? P.shape[1] : P.C;
    const int c = F.par & 1;
    double* rhsc = c ? F.rhs1 : P.rhsbuf;
    for (int i = threadIdx.x; i < R; i += FG_NT) rhsc[i] = P.T[(size_t)i * P.ld + (C - 1)];
    if (threadIdx.x == 0) {
        DevState x = init[k];
        x.r = -1; x.q = -1; x.qn = -1; x.pad[2] = 1; x.pad[3] = 0;
        F.rec[c] = x;
        x.pad[2] = 0;
        F.rec[c ^ 1] = x;
    }
}

// the latest record of every node (larger launch count) for the host and for the handle's own state record; which index it
// was goes to `cur` (the node's F.par of its next run)
__global__ __launch_bounds__(64) void lpx_group_fused_gather(const FusedParams* __restrict__ arr, DevState* __restrict__ out, int* __restrict__ cur)
{
    const FusedParams F = arr[blockIdx.x];
    const int which = F.rec[1].pad[2] > F.rec[0].pad[2] ? 1 : 0;
    const int32_t* s = reinterpret_cast<const int32_t*>(F.rec + which);
    int32_t* d = reinterpret_cast<int32_t*>(out + blockIdx.x);
    int32_t* d2 = reinterpret_cast<int32_t*>(F.P.st);
    for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 64) { const int32_t v = s[k]; d[k] = v; d2[k] = v; }
    if (threadIdx.x == 0) cur[blockIdx.x] = which;
}

// device-side compaction record of a group (ints): the live list the update workgroups of launch L go by was written by the last
// select workgroup of launch L - 1 (or by the host for the first launch of a window), double-buffered on the launch parity
// layout: [parity 0: count, 15 pad, list[cap]] [parity 1: the same] [arrival counter of the select workgroups, 15 pad] [flags[cap]: "slot s goes on"]
static constexpr int FG_COMP_HDR = 16;
__host__ __device__ constexpr int fg_comp_region(int cap) { return FG_COMP_HDR + cap; }

template <bool NT>
__device__ __forceinline__ void lpx_group_fused_body(const FusedParams* __restrict__ arr, const int* __restrict__ live, int nlive,
                                                     int per_node, int lpar, int mixmod, const int* __restrict__ comp_rd, int* comp, int cap)
{
    __shared__ double s_v[FG_NT / 64];
    __shared__ int s_i[FG_NT / 64];
    const int t = threadIdx.x;
    const int bid = blockIdx.x;
    const bool is_select = bid < nlive;
    const int par = lpar & 1;
    int node, ublk = 0;
    if (is_select) node = __builtin_amdgcn_readfirstlane(live[bid]);
    else {
        // update workgroups take their node from the DEVICE's live list: nodes that finished in an earlier launch of this window have
        // left it, and the workgroups beyond the live ones (the tail of the grid) leave after one load
        // (comp_rd = this launch's parity region, read-only in this launch: scalar loads; the kernel writes the other region through `comp`)
        const int u = bid - nlive;
        const int n_dev = comp_rd[0];
        if (u >= n_dev * per_node) return;
        node = comp_rd[FG_COMP_HDR + u / per_node];
        ublk = u % per_node;
    }
    const FusedParams F = arr[node];
    const SelParams& P = F.P;
    const int c = (lpar ^ F.par) & 1;
    // the record is workgroup-uniform but lives behind a pointer the kernel also writes through (index 1 - c), so the compiler
    // loads it into VECTOR registers: every field that is used goes through readfirstlane, and nothing keeps the struct alive
    const DevState* curp = F.rec + c;
    DevState* nxt = F.rec + (c ^ 1);
    const int status = __builtin_amdgcn_readfirstlane(curp->status);
    const int pr = __builtin_amdgcn_readfirstlane(curp->r);          // pending pivot row (-1: nothing to apply)
    const int seq = __builtin_amdgcn_readfirstlane(curp->pad[2]), buf = __builtin_amdgcn_readfirstlane(curp->pad[3]);
    const int R = __builtin_amdgcn_readfirstlane(P.shape ? P.shape[0] : P.R), C = __builtin_amdgcn_readfirstlane(P.shape ? P.shape[1] : P.C);
    const size_t ld = (size_t)P.ld;
    const double* __restrict__ src = buf ? F.T1 : P.T;
    double* __restrict__ dst = buf ? P.T : F.T1;
    const double* __restrict__ prowc = c ? F.prow1 : P.prow;
    const double* __restrict__ facc = c ? P.col1 : P.col0;

    if (!is_select) {
        // ---------------- update(k): T_{k+1} = T_k - fac (x) prow, row r replaced by the normalised pivot row ----------------
        if (status != LPX_RUNNING || pr < 0) return;
        const int ncw = (P.ld + 127) / 128;
        const int nunits = ncw * ((R + UPDS_ROWS - 1) / UPDS_ROWS);
        const int lane = t & 63;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
        const int unit = ublk * (FG_NT / 64) + wave;
        if (unit >= nunits) return;
        const int cw = unit % ncw, rb = unit / ncw;
        const int col = cw * 128 + lane * 2;
        if (col >= P.ld) return;
        const int row0 = rb * UPDS_ROWS;
        const LPX_GLOBAL double* gprow = (const LPX_GLOBAL double*)prowc;
        const LPX_GLOBAL double* gfac = (const LPX_GLOBAL double*)facc;
        const double2 p = upd_load_g<false>(gprow + col);
        const LPX_GLOBAL double* sb = (const LPX_GLOBAL double*)src + (size_t)row0 * ld + col;
        LPX_GLOBAL double* db = (LPX_GLOBAL double*)dst + (size_t)row0 * ld + col;
        if (row0 + UPDS_ROWS <= R && (pr < row0 || pr >= row0 + UPDS_ROWS)) {
            double2 v[UPDS_ROWS];
            double f[UPDS_ROWS];
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) v[k] = upd_load_g<NT>(sb + (size_t)k * ld);
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) f[k] = gfac[row0 + k];
#pragma unroll
            for (int k = 0; k < UPDS_ROWS; ++k) {
                v[k].x = v[k].x - f[k] * p.x;       // mul, then sub: contraction is off
                v[k].y = v[k].y - f[k] * p.y;
            }
            if (NT && mixmod > 0 && (mixmod == 1 || rb % mixmod == 0)) {
#pragma unroll
                for (int k = 0; k < UPDS_ROWS - 1; ++k) upd_store_g<true>(db + (size_t)k * ld, v[k]);
                upd_store_g<false>(db + (size_t)(UPDS_ROWS - 1) * ld, v[UPDS_ROWS - 1]);
            } else {
#pragma unroll
                for (int k = 0; k < UPDS_ROWS; ++k) upd_store_g<NT>(db + (size_t)k * ld, v[k]);
            }
            return;
        }
#pragma unroll 1
        for (int k = 0; k < UPDS_ROWS; ++k) {
            const int i = row0 + k;
            if (i >= R) break;
            double2 o = p;                                   // row r: the normalised pivot row
            if (i != pr) {
                const double2 v = upd_load_g<NT>(sb + (size_t)k * ld);
                const double f = gfac[i];
                o.x = v.x - f * p.x;
                o.y = v.y - f * p.y;
            }
            upd_store_g<NT>(db + (size_t)k * ld, o);
        }
        return;
    }

    // ---------------- select(k+1) on T_{k+1}, read as T_k with pivot k's correction ----------------
    auto select = [&]() -> int {        // returns 1 while the node goes on (a pivot is pending for the next launch)
    const int nbuf = pr >= 0 ? (buf ^ 1) : buf;              // where T_{k+1} lives once this launch is over
    if (status != LPX_RUNNING) {
        if (t == 0) { DevState x = *curp; x.pad[2] = seq + 1; *nxt = x; }
        return 0;
    }
    double* __restrict__ prown = c ? P.prow : F.prow1;
    double* __restrict__ facn = c ? P.col0 : P.col1;
    const double* __restrict__ rhsc = c ? F.rhs1 : P.rhsbuf;
    double* __restrict__ rhsn = c ? P.rhsbuf : F.rhs1;
    const int m = R - 1, rhs = C - 1;
    const size_t mx = (size_t)(P.R > P.C ? P.R : P.C);
    double* zrow = P.ws;                                     // objective row of T_{k+1}
    double* lrow = P.ws + mx;                                // row r of T_{k+1}
    double* rat = P.ws + 2 * mx;                             // ratios of the scan at hand
    const double inf = __builtin_inf();
    {   // RHS column and objective row of T_{k+1}, as the update stores them (mul, then sub; row r: the normalised pivot row)
        const double prhs = pr >= 0 ? prowc[rhs] : 0.0;
        for (int i = t; i < R; i += FG_NT) {
            const double h = rhsc[i];
            double v = h;
            if (pr >= 0) { const double u = h - facc[i] * prhs; v = (i == pr) ? prhs : u; }
            rhsn[i] = v;
        }
        const double fm = pr >= 0 ? facc[m] : 0.0;
        const double* orow = src + (size_t)m * ld;
        for (int j = t; j < C; j += FG_NT) {
            const double z = orow[j];
            zrow[j] = pr >= 0 ? z - fm * prowc[j] : z;
        }
    }
    __syncthreads();
    // column q of T_{k+1} -> the factors of pivot k+1; with `ratios`: ChooseLeaving's ratios rhs_i / a_i (a_i > eps)
    auto column = [&](int q, bool ratios) {
        const double pq = pr >= 0 ? prowc[q] : 0.0;
        for (int i = t; i < R; i += FG_NT) {
            const double v = src[(size_t)i * ld + q];
            double a = v;
            if (pr >= 0) { const double u = v - facc[i] * pq; a = (i == pr) ? pq : u; }
            facn[i] = a;
            if (ratios && i < m) rat[i] = a > P.eps ? rhsn[i] / a : inf;
        }
    };
    // row r of T_{k+1}; with `ratios`: the dual loop's ratios z_j / (-a_j) (a_j < -eps), Models/DualSimplex.cs:79-91
    auto row = [&](int r, bool ratios) {
        const double fr = pr >= 0 ? facc[r] : 0.0;
        const double* trow = src + (size_t)r * ld;
        for (int j = t; j < C; j += FG_NT) {
            const double v = trow[j];
            double a = v;
            if (pr >= 0) { const double pc = prowc[j]; const double u = v - fr * pc; a = (r == pr) ? pc : u; }
            lrow[j] = a;
            if (ratios && j < rhs) rat[j] = a < -P.eps ? zrow[j] / (-a) : inf;
        }
    };
    int phase = __builtin_amdgcn_readfirstlane(curp->phase);
    const int fdf_count = __builtin_amdgcn_readfirstlane(curp->fdf_count), dual_iter = __builtin_amdgcn_readfirstlane(curp->dual_iter);
    const int primal_count = __builtin_amdgcn_readfirstlane(curp->primal_count), iter = __builtin_amdgcn_readfirstlane(curp->iter);
    int r = -1, q = -1;
    int final_status = LPX_RUNNING;
    bool have_row = false;
    // ChooseEntering's column (first strict minimum of the objective row below -eps, Models/PrimalSimplex.cs:205-220) and the dual
    // loop's leaving row (most negative RHS, Models/DualSimplex.cs:45-55) depend on T_{k+1} alone, not on the phase: both up front,
    // so that the state machine below has ONE scan site (inlined once: its 16-ratio register block is what the kernel's
    // register count -- shared with the update waves -- can afford)
    const int qz = block_first_min_below<FG_NT>(zrow, 1, rhs, P.eps, s_v, s_i);
    const int rr = block_first_min_below<FG_NT>(rhsn, 1, m, P.eps, s_v, s_i);
    // state machine: ForceDualFeasibility -> dual loop -> (repaired mode) primal clean-up; the same hops as lpx_select_body
    for (int hop = 0; hop < 3 && final_status == LPX_RUNNING && r < 0; ++hop) {
        int L; double tol;
        if (phase == 0) {
            if (fdf_count >= P.fdf_guard || qz < 0) { phase = 1; continue; }
            q = qz; column(q, true); L = m; tol = P.tol_fdf;
        } else if (phase == 1) {
            if (dual_iter >= P.max_iter) { final_status = LPX_ITER_LIMIT; break; }
            if (rr < 0) {
                if (P.cleanup && qz >= 0) { phase = 2; continue; }
                final_status = LPX_OPTIMAL; break;
            }
            row(rr, true); have_row = true; L = rhs; tol = P.tol_dual;
        } else {
            if (primal_count >= P.max_iter - dual_iter) { final_status = LPX_ITER_LIMIT; break; }
            if (qz < 0) { final_status = LPX_OPTIMAL; break; }
            q = qz; column(q, true); L = m; tol = P.tol_primal;
        }
        __syncthreads();                                     // the ratios are complete (and visible: same CU)
        const int w = block_hysteresis_segments<FG_NT / 64>(L, tol, CompactRatio{rat});
        __syncthreads();                                     // segment records and `rat` may be reused by the next scan
        if (phase == 1) {
            if (w < 0) { final_status = LPX_INFEASIBLE; break; }
            r = rr; q = w;
            column(q, false);
        } else if (w < 0) {
            q = -1;
            if (phase == 0) { phase = 1; continue; }
            final_status = LPX_UNBOUNDED; break;
        } else r = w;
    }
    if (final_status != LPX_RUNNING || r < 0) {
        if (t == 0) {
            DevState x = *curp;
            x.status = (final_status == LPX_RUNNING) ? LPX_OPTIMAL : final_status;
            x.phase = phase; x.r = -1; x.q = -1; x.qn = -1; x.pad[2] = seq + 1; x.pad[3] = nbuf;
            *nxt = x;
        }
        return 0;
    }
    // pivot prep (Models/PrimalSimplex.cs:249-250): the normalised pivot row of pivot k+1, true division
    if (!have_row) row(r, false);
    __syncthreads();
    const double piv = lrow[q];
    for (int j = t; j < C; j += FG_NT) prown[j] = lrow[j] / piv;
    if (t == 0) {
        P.basis[r] = q;                                          // basis[leaving] = entering, :110
        if (iter < P.trace_cap) { P.trace[2 * iter] = r; P.trace[2 * iter + 1] = q; }
        DevState x = *curp;
        x.status = LPX_RUNNING; x.iter = iter + 1; x.r = r; x.q = q; x.phase = phase; x.qn = -1;
        if (phase == 0) x.fdf_count = fdf_count + 1;
        else if (phase == 1) x.dual_iter = dual_iter + 1;
        else x.primal_count = primal_count + 1;
        x.pad[2] = seq + 1; x.pad[3] = nbuf;
        *nxt = x;
    }
    return 1;
    };
    const int alive = select();
    // ---- compaction for the NEXT launch: the last select workgroup to get here writes the list of the nodes that go on.  Hand-off as in
    //      lpx_select_mb: an agent-scope store of this slot's flag, wait for it, ONE agent-scope add; the workgroup whose add returns
    //      nlive - 1 reads the flags with agent-scope loads.  Nothing of index `par` is written here; the update workgroups read only that.
    if (t < 64) {
        int* arrive = comp + 2 * fg_comp_region(cap);
        int* flags = arrive + FG_COMP_HDR;
        int last = 0;
        if (t == 0) {
            __hip_atomic_store(&flags[bid], alive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            last = (__hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nlive - 1) ? 1 : 0;
        }
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            int* outr = comp + (par ^ 1) * fg_comp_region(cap);
            int* out = outr + FG_COMP_HDR;
            int k = 0;
            for (int s0 = 0; s0 < nlive; s0 += 64) {
                const int sidx = s0 + t;
                const int f = sidx < nlive ? __hip_atomic_load(&flags[sidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                const unsigned long long mk = __ballot(f != 0);
                if (f) out[k + __popcll(mk & ((1ull << t) - 1ull))] = live[sidx];
                k += __popcll(mk);
            }
            if (t == 0) {
                outr[0] = k;
                __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ __launch_bounds__(FG_NT) __attribute__((amdgpu_waves_per_eu(6))) void lpx_group_fused(const FusedParams* arr, const int* live, int nlive, int per_node, int lpar, int mixmod, const int* comp_rd, int* comp, int cap)
{ lpx_group_fused_body<true>(arr, live, nlive, per_node, lpar, mixmod, comp_rd, comp, cap); }
__global__ __launch_bounds__(FG_NT) __attribute__((amdgpu_waves_per_eu(6))) void lpx_group_fused_c(const FusedParams* arr, const int* live, int nlive, int per_node, int lpar, int mixmod, const int* comp_rd, int* comp, int cap)
{ lpx_group_fused_body<false>(arr, live, nlive, per_node, lpar, mixmod, comp_rd, comp, cap); }

hipError_t launch_group_fused_init(const FusedParams* arr, const int* fresh, int nfresh, const DevState* init, hipStream_t s)
{
    if (nfresh <= 0) return hipSuccess;
    hipLaunchKernelGGL(lpx_group_fused_init, dim3(nfresh), dim3(FG_NT), 0, s, arr, fresh, init);
    return hipGetLastError();
}
hipError_t launch_group_fused_gather(const FusedParams* arr, int count, DevState* out, int* cur, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_group_fused_gather, dim3(count), dim3(64), 0, s, arr, out, cur);
    return hipGetLastError();
}
// workgroups of the update part for a node of capacity (ld, R): four waves each
int group_fused_blocks(int ld, int R)
{
    const int nunits = ((ld + 127) / 128) * ((R + UPDS_ROWS - 1) / UPDS_ROWS);
    return (nunits + (FG_NT / 64) - 1) / (FG_NT / 64);
}
// live_bytes: tableau bytes of the live nodes (one buffer each): both buffers of the group at home in the Infinity Cache ->
// default policy; beyond that nontemporal loads and the mixed store policy of lpx_pivot_fused
int group_fused_comp_ints(int cap) { return 3 * fg_comp_region(cap); }
int group_fused_comp_hdr() { return FG_COMP_HDR; }
hipError_t launch_group_fused(const FusedParams* arr, const int* live, int nlive, int per_node, int lpar, size_t live_bytes, hipStream_t s,
                              int* comp, int cap, hipEvent_t e0, hipEvent_t e1)
{
    if (nlive <= 0) return hipSuccess;
    static const int forced = [] { const char* e = std::getenv("LPX_UPDATE_POLICY"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1; }();
    int pol = live_bytes <= FUSED_CACHED_BYTES ? 0 : (live_bytes <= UPD_MIXED_BYTES ? 2 : 1);
    if (forced >= 0) pol = forced;
    static const int mm_forced = [] { const char* e = std::getenv("LPX_UPDATE_MIXMOD"); return e ? std::atoi(e) : 0; }();
    const int mixmod = pol == 2 ? (mm_forced > 0 ? mm_forced : (int)((live_bytes + UPD_MIX_STEP_BYTES - 1) / UPD_MIX_STEP_BYTES)) : 0;
    auto kern = pol == 0 ? lpx_group_fused_c : lpx_group_fused;
    const unsigned nblocks = (unsigned)nlive * (unsigned)(1 + per_node);
    const int* comp_rd = comp + (lpar & 1) * fg_comp_region(cap);
    if (e0 && e1) hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(FG_NT), 0, s, e0, e1, 0, arr, live, nlive, per_node, lpar & 1, mixmod, comp_rd, comp, cap);
    else hipLaunchKernelGGL(kern, dim3(nblocks), dim3(FG_NT), 0, s, arr, live, nlive, per_node, lpar & 1, mixmod, comp_rd, comp, cap);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#ifdef LPX_STAMPS
hipError_t debug_copy_stamps(unsigned long long* out, int clear)
{
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(lpx_g_stamps), sizeof(unsigned long long) * 32);
    if (e == hipSuccess && clear) { unsigned long long z[32] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(lpx_g_stamps), z, sizeof(z)); }
    return e;
}
#endif

static constexpr int SEL_LDS_MAX_DOUBLES = 16 * 1024;          // 128 KB of ratios: tableaux up to 16k rows / columns

hipError_t kernels_init()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lpx_select), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * SEL_LDS_MAX_DOUBLES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(lpx_select_b), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * SEL_LDS_MAX_DOUBLES);
    return e;
}

// doubles of dynamic LDS the dual select needs for a tableau of capacity R x C (0 = too long, scan from global memory)
static int select_lds_doubles(int R, int C)
{
    const int L = ((R > C ? R : C) + 1) & ~1;
    return L <= SEL_LDS_MAX_DOUBLES ? L : 0;
}

hipError_t launch_select(const SelParams& p, hipStream_t s)
{
    const int L = select_lds_doubles(p.R, p.C);
    hipLaunchKernelGGL(lpx_select, dim3(1), dim3(SEL_NT), sizeof(double) * L, s, p, L);
    return hipGetLastError();
}

hipError_t launch_select_la(const SelParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_select_la, dim3(1), dim3(SEL_NT), 0, s, p);
    return hipGetLastError();
}

int select_mb_blocks(int C)
{
    int b = (C + MB_NT - 1) / MB_NT;          // at least one column per lane and pass
    if (b < 1) b = 1;
    if (b > 32) b = 32;
    return b;
}

hipError_t launch_select_mb(const SelParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_select_mb, dim3(p.nblk), dim3(MB_NT), 0, s, p);
    return hipGetLastError();
}

// which streaming form a tableau of `bytes` takes: 0 = none (it lives in the Infinity Cache), 2 = mixed store policy, 1 = all nt
int update_policy(int ld, int R)
{
    // LPX_UPDATE_POLICY=0|1|2 forces one form (diagnostic: tools/probe_policy.py measures the three on one tableau)
    static const int forced = [] { const char* e = std::getenv("LPX_UPDATE_POLICY"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1; }();
    if (forced >= 0) return forced;
    const size_t bytes = sizeof(double) * (size_t)ld * (size_t)R;
    if (bytes <= UPD_STREAM_BYTES) return 0;
    return bytes <= UPD_MIXED_BYTES ? 2 : 1;
}
// every `mixmod`-th row block of the mixed form keeps one row in three in the cache: about a cache-full of the tableau in all
// (256 MiB at 768 MiB -> every block up to there, every second block up to 1.5 GiB, ...)
static int update_mixmod(int ld, int R)
{
    static const int forced = [] { const char* e = std::getenv("LPX_UPDATE_MIXMOD"); return e ? std::atoi(e) : 0; }();   // diagnostic
    if (forced > 0) return forced;
    const size_t bytes = sizeof(double) * (size_t)ld * (size_t)R;
    return (int)((bytes + UPD_MIX_STEP_BYTES - 1) / UPD_MIX_STEP_BYTES);
}

hipError_t launch_update_mb(const SelParams& p, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    // the streaming forms read the entering column select's last workgroup reduced: both sides follow SelParams::qsel
    const int pol = !p.qsel ? 0 : (update_policy(p.ld, p.R) == 2 ? 2 : 1);
    const int rows = pol ? UPDS_ROWS : UPD_ROWS, nth = pol ? UPDS_NT : UPD_NT;
    const int ncw = (p.ld + 127) / 128, nunits = ncw * ((p.R + rows - 1) / rows);
    const int nblocks = (nunits + (nth / 64) - 1) / (nth / 64);
    const int forced = p.mode == MODE_FORCED ? 1 : 0;
    if (pol == 0) {
        if (e0 && e1)
            hipExtLaunchKernelGGL(lpx_update_mb, dim3(nblocks), dim3(nth), 0, s, e0, e1, 0, p.T, p.ld, p.R, p.C, p.shape,
                                  (const double*)p.prow, p.col0, p.col1, p.rhsbuf, (const DevState*)p.st, p.us,
                                  (const double*)p.part_v, (const int32_t*)p.part_i, p.nblk, forced, ncw, nunits);
        else
            hipLaunchKernelGGL(lpx_update_mb, dim3(nblocks), dim3(nth), 0, s, p.T, p.ld, p.R, p.C, p.shape,
                               (const double*)p.prow, p.col0, p.col1, p.rhsbuf, (const DevState*)p.st, p.us,
                               (const double*)p.part_v, (const int32_t*)p.part_i, p.nblk, forced, ncw, nunits);
        return hipGetLastError();
    }
    auto kern = pol == 2 ? lpx_update_mb_m : lpx_update_mb_s;
    const int mixmod = update_mixmod(p.ld, p.R);
    if (e0 && e1)
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(nth), 0, s, e0, e1, 0, p.T, p.ld, p.R, p.C, p.shape,
                              (const double*)p.prow, p.col0, p.col1, p.rhsbuf, (const DevState*)p.st, p.us,
                              (const double*)p.part_v, (const int32_t*)p.part_i, p.nblk, forced, ncw, nunits, mixmod);
    else
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(nth), 0, s, p.T, p.ld, p.R, p.C, p.shape,
                           (const double*)p.prow, p.col0, p.col1, p.rhsbuf, (const DevState*)p.st, p.us,
                           (const double*)p.part_v, (const int32_t*)p.part_i, p.nblk, forced, ncw, nunits, mixmod);
    return hipGetLastError();
}

// one iteration of a whole group: `arr` holds `count` parameter records in device memory
hipError_t launch_group_iter(const SelParams* arr, int count, int dual, int max_nblk, int max_upd_blocks, hipStream_t s, int maxR, int maxC)
{
    if (dual) {
        const int L = select_lds_doubles(maxR, maxC);
        hipLaunchKernelGGL(lpx_select_b, dim3(1, count), dim3(SEL_NT), sizeof(double) * L, s, arr, L);
        hipLaunchKernelGGL(lpx_update_b, dim3(max_upd_blocks, count), dim3(UPD_NT), 0, s, arr);
    } else {
        hipLaunchKernelGGL(lpx_select_mb_b, dim3(max_nblk, count), dim3(MB_NT), 0, s, arr);
        hipLaunchKernelGGL(lpx_update_mb_b, dim3(max_upd_blocks, count), dim3(UPD_NT), 0, s, arr);
    }
    return hipGetLastError();
}
// parent parking for a whole group: every finished node's tableau and basis into its store slot, one launch
__global__ __launch_bounds__(256) void lpx_park_many(const ParkDesc* __restrict__ descs)
{
    const ParkDesc D = descs[blockIdx.y];
    const size_t n2 = D.doubles / 2;                                  // leading dimensions are multiples of 16: whole double2s
    const double2* __restrict__ src = reinterpret_cast<const double2*>(D.srcT);
    double2* __restrict__ dst = reinterpret_cast<double2*>(D.dstT);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < D.m; i += 256) D.dstB[i] = D.srcB[i];
}
hipError_t launch_park_many(const ParkDesc* descs, int count, int blocks_per_node, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_park_many, dim3(blocks_per_node, count), dim3(256), 0, s, descs);
    return hipGetLastError();
}

// state records of a whole group in one launch each way (pinned host array <-> every node's device record): a group of 64
// nodes paid 64 small copies per begin and per poll
__global__ __launch_bounds__(64) void lpx_states_scatter(const SelParams* __restrict__ arr, const DevState* __restrict__ src)
{
    const int32_t* s = reinterpret_cast<const int32_t*>(src + blockIdx.x);
    int32_t* d = reinterpret_cast<int32_t*>(arr[blockIdx.x].st);
    for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 64) d[k] = s[k];
}
__global__ __launch_bounds__(64) void lpx_states_gather(const SelParams* __restrict__ arr, DevState* __restrict__ dst)
{
    const int32_t* s = reinterpret_cast<const int32_t*>(arr[blockIdx.x].st);
    int32_t* d = reinterpret_cast<int32_t*>(dst + blockIdx.x);
    for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 64) d[k] = s[k];
}
hipError_t launch_states_scatter(const SelParams* arr, const DevState* src_pinned, int count, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_states_scatter, dim3(count), dim3(64), 0, s, arr, src_pinned);
    return hipGetLastError();
}
hipError_t launch_states_gather(const SelParams* arr, DevState* dst_pinned, int count, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_states_gather, dim3(count), dim3(64), 0, s, arr, dst_pinned);
    return hipGetLastError();
}
hipError_t launch_group_init(const SelParams* arr, int count, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_la_init_b, dim3(1, count), dim3(SEL_NT), 0, s, arr);
    return hipGetLastError();
}
hipError_t launch_group_rhs_init(const SelParams* arr, int count, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_rhs_init_b, dim3(1, count), dim3(SEL_NT), 0, s, arr);
    return hipGetLastError();
}
hipError_t launch_rhs_init(const SelParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_rhs_init, dim3(1), dim3(SEL_NT), 0, s, p);
    return hipGetLastError();
}
int update_blocks(int ld, int R)
{
    const int nunits = ((ld + 127) / 128) * ((R + UPD_ROWS - 1) / UPD_ROWS);
    return (nunits + (UPD_NT / 64) - 1) / (UPD_NT / 64);
}

hipError_t launch_la_init(const SelParams& p, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_la_init, dim3(1), dim3(SEL_NT), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_update(double* T, int ld, int R, int C, const int32_t* shape, const double* prow, double* fac0, double* fac1,
                         double* rhsbuf, const DevState* st, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    const int pol = update_policy(ld, R);
    const int rows = pol ? UPDS_ROWS : UPD_ROWS, nth = pol ? UPDS_NT : UPD_NT;
    const int ncw = (ld + 127) / 128, nunits = ncw * ((R + rows - 1) / rows);
    const int nblocks = (nunits + (nth / 64) - 1) / (nth / 64);
    if (pol == 0) {
        if (e0 && e1)
            hipExtLaunchKernelGGL(lpx_update, dim3(nblocks), dim3(nth), 0, s, e0, e1, 0, T, ld, R, C, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits);
        else
            hipLaunchKernelGGL(lpx_update, dim3(nblocks), dim3(nth), 0, s, T, ld, R, C, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits);
        return hipGetLastError();
    }
    auto kern = pol == 2 ? lpx_update_m : lpx_update_s;
    const int mixmod = update_mixmod(ld, R);
    if (e0 && e1)
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(nth), 0, s, e0, e1, 0, T, ld, R, C, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits, mixmod);
    else
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(nth), 0, s, T, ld, R, C, shape, prow, fac0, fac1, rhsbuf, st, ncw, nunits, mixmod);
    return hipGetLastError();
}

// Cache policy of the fused launch.  Two buffers share the Infinity Cache, so the default policy only pays while BOTH fit with
// room to spare; beyond that the streaming mix wins at every size, well below the in-place kernels' own crossover
// (tools/probe_fused_mid.py, us per pivot, two-launch in place / fused default policy / fused streaming mix):
//    57 MB 24.1 / 21.4 / 22.8     101 MB 37.8 / 30.5 / 34.3     157 MB 53.5 / 49.2 / 49.1     190 MB 64.5 / 65.5 / 58.3
//   227 MB 73.1 / 77.2 / 69.0     266 MB 86.1 / 91.1 / 80.3     308 MB 102.7 / 92.4 / 92.2    403 MB 128.3 / 120.0 / 120.2
int fused_policy(int ld, int R)
{
    static const int forced = [] { const char* e = std::getenv("LPX_UPDATE_POLICY"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : -1; }();
    if (forced >= 0) return forced;
    const size_t bytes = sizeof(double) * (size_t)ld * (size_t)R;
    if (bytes <= FUSED_CACHED_BYTES) return 0;
    return bytes <= UPD_MIXED_BYTES ? 2 : 1;
}

hipError_t launch_fused_init(const FusedParams& f, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_fused_init, dim3(1), dim3(SEL_NT), 0, s, f);
    return hipGetLastError();
}

hipError_t launch_pivot_fused(const FusedParams& f0, int par, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    FusedParams f = f0; f.par = par & 1;
    const int ld = f.P.ld, R = f.P.R;
    const int ncw = (ld + 127) / 128, nunits = ncw * ((R + UPDS_ROWS - 1) / UPDS_ROWS);
    const int nblocks = f.P.nblk + (nunits + (FP_NT / 64) - 1) / (FP_NT / 64);
    const int pol = fused_policy(ld, R);
    const int mixmod = pol == 2 ? update_mixmod(ld, R) : 0;      // 0: every store nontemporal
    auto kern = pol == 0 ? lpx_pivot_fused_c : lpx_pivot_fused;  // both buffers at home in the Infinity Cache: default policy
    if (e0 && e1)
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(FP_NT), 0, s, e0, e1, 0, f, ncw, nunits, mixmod);
    else
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(FP_NT), 0, s, f, ncw, nunits, mixmod);
    return hipGetLastError();
}

}  // namespace lpx
