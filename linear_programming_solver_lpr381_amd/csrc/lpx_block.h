// lpx_block.h -- workgroup-level device primitives (wave64) shared by the select kernels of the
// tableau and revised paths: (value,index) min-reduce, exclusive scans, and the exact parallel form
// of the reference's hysteresis ratio scan.
#pragma once
#include "lpx_internal.h"
#include <limits.h>

namespace lpx {

static constexpr int SEL_NT = 1024;          // lanes of the single-workgroup select kernels
static constexpr int SEL_NW = SEL_NT / 64;   // waves
static constexpr int MB_NT = 256;            // lanes of one workgroup of the multi-workgroup select
// words of the partial-index buffer behind its 128 entries, each on a 128-byte line of its own: the next entering column
// reduced by the last select workgroup, and the arrival counter of the select workgroups
static constexpr int MB_QREC = 128;
static constexpr int MB_CNT = 160;
static constexpr int MB_MAXB = 64;           // at most this many workgroups (partials fit one wave)

// ------------------------------------------------------------------------------------------------
// workgroup primitives (wave64)
// ------------------------------------------------------------------------------------------------
struct MinIdx { double v; int i; };

__device__ __forceinline__ MinIdx mi_pick(MinIdx a, MinIdx b)
{
    // strict minimum, lowest index on equal values ("first index of the strict minimum")
    if (b.v < a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// DPP cross-lane moves (row_shr within 16-lane rows, row_bcast15/31 across rows): a few cycles each,
// where __shfl lowers to ds_bpermute_b32 through the LDS crossbar (~100+ cycles on a dependent chain).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double identity, double x)
{
    int lo = __builtin_amdgcn_update_dpp(__double2loint(identity), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(identity), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int identity, int x)
{
    return __builtin_amdgcn_update_dpp(identity, x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ double fmin_nn(double a, double b) { return b < a ? b : a; }

// wave-wide minimum of a double, returned in every lane (inclusive scan to lane 63 + readlane)
__device__ __forceinline__ double wave_min_f64(double x)
{
    const double inf = __builtin_inf();
    x = fmin_nn(x, dpp_f64<0x111, 0xf>(inf, x));     // row_shr:1
    x = fmin_nn(x, dpp_f64<0x112, 0xf>(inf, x));     // row_shr:2
    x = fmin_nn(x, dpp_f64<0x114, 0xf>(inf, x));     // row_shr:4
    x = fmin_nn(x, dpp_f64<0x118, 0xf>(inf, x));     // row_shr:8
    x = fmin_nn(x, dpp_f64<0x142, 0xa>(inf, x));     // row_bcast:15 into rows 1 and 3
    x = fmin_nn(x, dpp_f64<0x143, 0xc>(inf, x));     // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_min_i32(int x)
{
    const int id = INT_MAX;
    x = min(x, dpp_i32<0x111, 0xf>(id, x));
    x = min(x, dpp_i32<0x112, 0xf>(id, x));
    x = min(x, dpp_i32<0x114, 0xf>(id, x));
    x = min(x, dpp_i32<0x118, 0xf>(id, x));
    x = min(x, dpp_i32<0x142, 0xa>(id, x));
    x = min(x, dpp_i32<0x143, 0xc>(id, x));
    return __builtin_amdgcn_readlane(x, 63);
}

// (value, index) minimum with lowest index on equal values, in every lane.  Values are never NaN here
// (they only ever enter a MinIdx through a `<` comparison).
__device__ __forceinline__ MinIdx wave_min_idx(MinIdx x)
{
    MinIdx r;
    r.v = wave_min_f64(x.v);
    r.i = wave_min_i32(x.v == r.v ? x.i : INT_MAX);
    return r;
}

// All NT lanes call this. Returns the block-wide winner in every lane.
template <int NT = SEL_NT>
__device__ inline MinIdx block_min_idx(MinIdx x, double* s_v, int* s_i)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = wave_min_idx(x);
    __syncthreads();                     // protect s_v/s_i reuse
    if (lane == 0) { s_v[wave] = x.v; s_i[wave] = x.i; }
    __syncthreads();
    MinIdx y;
    y.v = s_v[lane & (NW - 1)];
    y.i = s_i[lane & (NW - 1)];
    y = wave_min_idx(y);
    return y;
}

// exclusive prefix-minimum over lanes in thread order; identity = +inf
template <int NT = SEL_NT>
__device__ inline double block_excl_scan_min(double x, double* s_v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double y = __shfl_up(inc, d, 64);
        if (lane >= d && y < inc) inc = y;
    }
    __syncthreads();
    if (lane == 63) s_v[wave] = inc;
    __syncthreads();
    double pre = __builtin_inf();        // minimum over all earlier waves
    for (int w = 0; w < wave; ++w) { double y = s_v[w]; if (y < pre) pre = y; }
    double up = __shfl_up(inc, 1, 64);   // minimum over earlier lanes of this wave
    if (lane == 0) up = __builtin_inf();
    return up < pre ? up : pre;
}

// exclusive prefix-sum over lanes in thread order; *total receives the block sum
template <int NT = SEL_NT>
__device__ inline int block_excl_scan_sum(int x, int* s_i, int* total)
{
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(inc, d, 64);
        if (lane >= d) inc += y;
    }
    __syncthreads();
    if (lane == 63) s_i[wave] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int w = 0; w < NW; ++w) { int y = s_i[w]; if (w < wave) pre += y; tot += y; }
    *total = tot;
    return pre + inc - x;
}

// ------------------------------------------------------------------------------------------------
// The reference's ratio scans are NOT argmins: `if (ratio < best - tol) { best = ratio; row = i; }`
// (Models/PrimalSimplex.cs:229-241, Models/DualSimplex.cs:79-91,:214-222, RevisedPrimalSimplex.cs:101-111)
// is a sequential hysteresis chain whose winner depends on the order of the rows.
//
// Exact wave-parallel form, no LDS and no barrier: the accepted rows are found one after the other as
// "the first position after the last accepted one whose ratio is below best - tol".  One wave keeps a
// segment of 64 x WH_PER ratios in registers (lane-strided, so the loads are fully coalesced); the
// search for the next accepted position is WH_PER wave-uniform ballots, and the number of rounds is the
// number of accepted rows (about ln L for random data, 1-2 for the degenerate ties of 0/1 programs).
// Segments are processed in order with (best, position) carried across.  `src` supplies den(k), num(k)
// -- plain loads, issued unconditionally and up front so that no load waits behind the eligibility
// test of another -- and value(den, num), which returns +inf for ineligible entries.
// ------------------------------------------------------------------------------------------------
static constexpr int WH_PER = 16;
#ifdef LPX_STAMPS
__device__ unsigned long long lpx_g_stamps[32];
#define LPX_HS(slot) do { if (threadIdx.x == 0 && blockIdx.x == 1) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); lpx_g_stamps[(slot)] += n_ - hs_prev_; hs_prev_ = n_; } } while (0)
#define LPX_HS_BEGIN unsigned long long hs_prev_ = __builtin_amdgcn_s_memtime();
#else
#define LPX_HS(slot) do {} while (0)
#define LPX_HS_BEGIN
#endif

// FLY = true: sources in LDS -- each ratio is formed as soon as its two operands are read, so only the 16 ratios
// stay live (32 VGPRs instead of 96; the resident kernels have 128 per lane and spill otherwise).
// (seg0, best0, win0): continue a scan whose segments before row seg0 ended with the carried (best, position).
// PER: ratios per lane and segment (default WH_PER = 16).  The chain is exact for any segment length (best and position are carried
// across segments); a kernel that is short of registers (lpx_resident_regs.hip: the tableau tile lives in VGPRs) scans with PER = 4.
template <class Src, bool FLY = false, int PER = 16>
__device__ __forceinline__ int wave_hysteresis_argmin(int L, double tol, const Src& src,
                                                      int seg0 = 0, double best0 = __builtin_inf(), int win0 = -1)
{
    constexpr int WH_PER = PER;
    const int lane = threadIdx.x & 63;
    double best = best0;
    int win = win0;
    LPX_HS_BEGIN
    for (int seg = seg0; seg < L; seg += 64 * WH_PER) {
        double rt[WH_PER];
        if constexpr (FLY) {
#pragma unroll
            for (int u = 0; u < WH_PER; ++u) {
                const int k = seg + u * 64 + lane;
                const int kc = min(k, L - 1);
                const double v = src.value(src.den(kc), src.num(kc));
                rt[u] = (k < L) ? v : __builtin_inf();
                __builtin_amdgcn_sched_barrier(0);      // one ratio (and its division) at a time: keeps the live set small
            }
        } else {
        double den[WH_PER], num[WH_PER];
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) {
            // clamped index instead of a guard: a guarded load becomes its own exec-masked branch with
            // its own s_waitcnt and the 32 loads serialise (measured: 14.8k cycles -> see DESIGN.md)
            const int k = min(seg + u * 64 + lane, L - 1);
            den[u] = src.den(k); num[u] = src.num(k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        LPX_HS(0);
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) {
            const int k = seg + u * 64 + lane;
            rt[u] = (k < L) ? src.value(den[u], num[u]) : __builtin_inf();
        }
        }
        LPX_HS(1);
        // Fast path.  Let rmin be the segment minimum and i* its first position.  No row of the segment
        // can be accepted unless rmin < fl(best - tol) (every ratio is >= rmin).  If that holds and every
        // OTHER row j has fl(r_j - tol) > rmin, then i* is accepted when the scan reaches it (whatever
        // was accepted before is either the carried best or some r_j, both leave rmin below the
        // threshold) and nothing after it can be (needs r < fl(rmin - tol) <= rmin).  So the chain ends
        // this segment at (rmin, i*) -- no replay needed.  Exact ties and near-ties fall through.
        MinIdx lm; lm.v = __builtin_inf(); lm.i = INT_MAX;
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) if (rt[u] < lm.v) { lm.v = rt[u]; lm.i = u * 64 + lane; }
        lm = wave_min_idx(lm);
        LPX_HS(2);
        if (!(lm.v < best - tol)) continue;              // nothing in this segment beats the carried best
        int inband = 0;
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) inband += ((rt[u] - tol) <= lm.v) ? 1 : 0;
        if (__ballot(inband >= 2) == 0ull && __popcll(__ballot(inband == 1)) == 1) {
            best = lm.v; win = seg + lm.i;
            LPX_HS(3);
            continue;
        }
        LPX_HS(4);
        int pos = -1;                                   // last accepted position inside this segment
        for (;;) {
            const double thr = best - tol;
            int found = -1;
#pragma unroll
            for (int u = 0; u < WH_PER; ++u) {
                if (found < 0) {
                    unsigned long long mk = __ballot(rt[u] < thr);
                    const int base = u * 64;
                    if (base + 63 <= pos) mk = 0ull;
                    else if (base <= pos) mk &= ~((2ull << (pos - base)) - 1ull);
                    if (mk) found = base + __ffsll((long long)mk) - 1;
                }
            }
            if (found < 0) break;
            double v = 0.0;
#pragma unroll
            for (int u = 0; u < WH_PER; ++u) if (u == (found >> 6)) v = rt[u];
            best = __shfl(v, found & 63, 64);
            win = seg + found;
            pos = found;
        }
        LPX_HS(5);
    }
    return win;
}

// The same scan with the segments of a long vector spread over the NW waves of a workgroup (multi-workgroup select at
// m > 1024: one wave alone walks ceil(m / 1024) segments one after the other, 2.5-3.5 us each).  Per segment only three
// things enter the chain -- its minimum, the first position of it, and whether that row is alone in the band
// [min, min + tol] -- and none of them depends on what was carried in, so the segments are evaluated in parallel and the
// chain is replayed over the <= 64 segment records by every lane.  A segment that would be entered through a tie / near-tie
// hands over to the exact sequential scan from that segment on (carried best and position included): same winner, always.
// All NW*64 lanes call this; one barrier.
// WAVE0_REPLAY: the exact replay after a tie runs on wave 0 only and is broadcast (single-workgroup kernels whose sources are
// strided gathers, and 0/1 programs whose dual ratios tie at 0 all the time: 16 waves replaying would multiply the traffic);
// false: every wave replays for itself, no second barrier (multi-workgroup select, where every wave did the whole scan before).
template <int NW, class Src, bool WAVE0_REPLAY = false>
__device__ __forceinline__ int block_hysteresis_segments(int L, double tol, const Src& src)
{
    constexpr int SEG = 64 * WH_PER;
    const int S = (L + SEG - 1) / SEG;
    if (S <= 1 || S > 64) return wave_hysteresis_argmin(L, tol, src);
    __shared__ double sg_v[64];
    __shared__ int sg_i[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int sgi = wave; sgi < S; sgi += NW) {
        const int seg = sgi * SEG;
        double den[WH_PER], num[WH_PER], rt[WH_PER];
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) {
            const int k = min(seg + u * 64 + lane, L - 1);
            den[u] = src.den(k); num[u] = src.num(k);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) {
            const int k = seg + u * 64 + lane;
            rt[u] = (k < L) ? src.value(den[u], num[u]) : __builtin_inf();
        }
        MinIdx lm; lm.v = __builtin_inf(); lm.i = INT_MAX;
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) if (rt[u] < lm.v) { lm.v = rt[u]; lm.i = u * 64 + lane; }
        lm = wave_min_idx(lm);
        int inband = 0;
#pragma unroll
        for (int u = 0; u < WH_PER; ++u) inband += ((rt[u] - tol) <= lm.v) ? 1 : 0;
        const bool clean = (__ballot(inband >= 2) == 0ull) && (__popcll(__ballot(inband == 1)) == 1);
        if (lane == 0) { sg_v[sgi] = lm.v; sg_i[sgi] = clean ? seg + lm.i : -1; }
    }
    __syncthreads();
    double best = __builtin_inf();
    int win = -1;
    for (int sgi = 0; sgi < S; ++sgi) {
        const double v = sg_v[sgi];
        if (!(v < best - tol)) continue;                 // nothing in this segment beats the carried best
        const int i = sg_i[sgi];
        if (i < 0) {                                     // ties: exact scan from here on (the decision is workgroup-uniform)
            if constexpr (!WAVE0_REPLAY) return wave_hysteresis_argmin(L, tol, src, sgi * SEG, best, win);
            __shared__ int sg_out;
            if (wave == 0) { const int w = wave_hysteresis_argmin(L, tol, src, sgi * SEG, best, win); if (lane == 0) sg_out = w; }
            __syncthreads();
            return sg_out;
        }
        best = v; win = i;
    }
    return win;
}

// wave 0 runs the scan, everybody gets the answer (single-workgroup select kernels)
template <class Src>
__device__ __forceinline__ int block_hysteresis_argmin(int L, double tol, const Src& src, int* s_out)
{
    if ((threadIdx.x >> 6) == 0) {
        const int w = wave_hysteresis_argmin(L, tol, src);
        if (threadIdx.x == 0) *s_out = w;
    }
    __syncthreads();
    const int r = *s_out;
    __syncthreads();                                    // s_out may be reused by the next scan
    return r;
}

// single-workgroup kernels: one segment -> wave 0 scans and the others wait (16 waves gathering the same strided column
// would only multiply the traffic); longer vectors -> the segments spread over all NW waves
template <int NW, class Src>
__device__ __forceinline__ int block_hysteresis_auto(int L, double tol, const Src& src, int* s_out)
{
    if (L <= 64 * WH_PER || L > 64 * 64 * WH_PER) return block_hysteresis_argmin(L, tol, src, s_out);
    const int r = block_hysteresis_segments<NW, Src, true>(L, tol, src);
    __syncthreads();                                    // its segment records may be reused by the next scan
    return r;
}

// ratio sources ------------------------------------------------------------------------------------
// rows: den = col[i*cs], num = rhs[i*rs], eligible den > eps, ratio = num/den
//       (ChooseLeaving, Models/PrimalSimplex.cs:229-241; revised ratio test, RevisedPrimalSimplex.cs:101-111)
struct RowRatio {
    const double* col; size_t cs; const double* rhs; size_t rs; double eps;
    __device__ __forceinline__ double den(int i) const { return col[(size_t)i * cs]; }
    __device__ __forceinline__ double num(int i) const { return rhs[(size_t)i * rs]; }
    __device__ __forceinline__ double value(double a, double b) const { return a > eps ? b / a : __builtin_inf(); }
};
// columns of the dual loop: den = T[r,j], num = T[m,j], eligible den < -eps, ratio = num/(-den)
//       (Models/DualSimplex.cs:79-91)
struct DualColRatio {
    const double* lrow; const double* zrow; double eps;
    __device__ __forceinline__ double den(int j) const { return lrow[j]; }
    __device__ __forceinline__ double num(int j) const { return zrow[j]; }
    __device__ __forceinline__ double value(double a, double z) const { return a < -eps ? z / (-a) : __builtin_inf(); }
};

// ChooseEntering, Models/PrimalSimplex.cs:205-220: first index of the strict minimum of `v[0..L)`
// (stride `stride`) below -eps, else -1.  Also used for the dual loop's leaving row (most negative
// RHS, Models/DualSimplex.cs:45-55).
template <int NT = SEL_NT>
__device__ inline int block_first_min_below(const double* v, size_t stride, int L, double eps,
                                     double* s_v, int* s_i)
{
    MinIdx m; m.v = -eps; m.i = INT_MAX;
    for (int j = threadIdx.x; j < L; j += NT) {
        double x = v[(size_t)j * stride];
        if (x < m.v) { m.v = x; m.i = j; }
    }
    m = block_min_idx<NT>(m, s_v, s_i);
    return m.i == INT_MAX ? -1 : m.i;
}


}  // namespace lpx
