// lpx_block.h -- workgroup-level device primitives (wave64) shared by the select kernels of the
// tableau and revised paths: (value,index) min-reduce, exclusive scans, and the exact parallel form
// of the reference's hysteresis ratio scan.
#pragma once
#include "lpx_internal.h"
#include <limits.h>

namespace lpx {

static constexpr int SEL_NT = 1024;          // lanes of the select workgroup
static constexpr int SEL_NW = SEL_NT / 64;   // waves
static constexpr int LIST_CAP = 2048;        // prefix-minimum records kept in LDS
static constexpr int SEL_LDS_DOUBLES = 16384;// 128 KiB of dynamic LDS for ratios

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

__device__ __forceinline__ MinIdx wave_min_idx(MinIdx x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        MinIdx y;
        y.v = __shfl_xor(x.v, d, 64);
        y.i = __shfl_xor(x.i, d, 64);
        x = mi_pick(x, y);
    }
    return x;
}

// All SEL_NT lanes call this. Returns the block-wide winner in every lane.
__device__ inline MinIdx block_min_idx(MinIdx x, double* s_v, int* s_i)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = wave_min_idx(x);
    __syncthreads();                     // protect s_v/s_i reuse
    if (lane == 0) { s_v[wave] = x.v; s_i[wave] = x.i; }
    __syncthreads();
    MinIdx y;
    y.v = s_v[lane & (SEL_NW - 1)];
    y.i = s_i[lane & (SEL_NW - 1)];
    y = wave_min_idx(y);
    return y;
}

// exclusive prefix-minimum over lanes in thread order; identity = +inf
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
__device__ inline int block_excl_scan_sum(int x, int* s_i, int* total)
{
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
    for (int w = 0; w < SEL_NW; ++w) { int y = s_i[w]; if (w < wave) pre += y; tot += y; }
    *total = tot;
    return pre + inc - x;
}

// ------------------------------------------------------------------------------------------------
// The reference's ratio scans are NOT argmins: `if (ratio < best - tol) { best = ratio; row = i; }`
// (Models/PrimalSimplex.cs:229-241, Models/DualSimplex.cs:79-91,:214-222) is a sequential hysteresis
// chain.  Exact parallel form: a candidate can only be accepted if it is a strict prefix-minimum
// record (accepted => ratio < fl(best - tol) <= min of all earlier ratios, because best never
// exceeds fl(prefix_min + tol) ... see DESIGN.md "hysteresis scan").  So: ratios -> LDS, exclusive
// prefix-min scan, compact the records in order, and one lane replays the chain over the (few)
// records.  Falls back to a plain sequential replay when the record list overflows.
// ratio(k) must return +inf for ineligible entries.
// ------------------------------------------------------------------------------------------------
template <class RatioFn>
__device__ int block_hysteresis_argmin(int L, double tol, RatioFn ratio, double* rbuf,
                                       int* s_list, double* s_v, int* s_i, int* s_out)
{
    const int t = threadIdx.x;
    const int per = (L + SEL_NT - 1) / SEL_NT;
    const int lo = t * per;
    const int hi = (lo + per < L) ? lo + per : L;
    double lmin = __builtin_inf();
    for (int k = lo; k < hi; ++k) {
        double r = ratio(k);
        rbuf[k] = r;
        if (r < lmin) lmin = r;
    }
    const double pre = block_excl_scan_min(lmin, s_v);
    int cnt = 0;
    double run = pre;
    for (int k = lo; k < hi; ++k) {
        double r = rbuf[k];
        if (r < run) { ++cnt; run = r; }
    }
    int total;
    int pos = block_excl_scan_sum(cnt, s_i, &total);
    if (total <= LIST_CAP) {
        run = pre;
        for (int k = lo; k < hi; ++k) {
            double r = rbuf[k];
            if (r < run) { s_list[pos++] = k; run = r; }
        }
    }
    __syncthreads();
    if (t == 0) {
        double best = __builtin_inf();
        int win = -1;
        if (total <= LIST_CAP) {
            for (int e = 0; e < total; ++e) {
                int k = s_list[e];
                double r = rbuf[k];
                if (r < best - tol) { best = r; win = k; }
            }
        } else {
            for (int k = 0; k < L; ++k) {
                double r = rbuf[k];
                if (r < best - tol) { best = r; win = k; }
            }
        }
        *s_out = win;
    }
    __syncthreads();
    return *s_out;
}

// ChooseEntering, Models/PrimalSimplex.cs:205-220: first index of the strict minimum of `v[0..L)`
// (stride `stride`) below -eps, else -1.  Also used for the dual loop's leaving row (most negative
// RHS, Models/DualSimplex.cs:45-55).
__device__ inline int block_first_min_below(const double* v, size_t stride, int L, double eps,
                                     double* s_v, int* s_i)
{
    MinIdx m; m.v = -eps; m.i = INT_MAX;
    for (int j = threadIdx.x; j < L; j += SEL_NT) {
        double x = v[(size_t)j * stride];
        if (x < m.v) { m.v = x; m.i = j; }
    }
    m = block_min_idx(m, s_v, s_i);
    return m.i == INT_MAX ? -1 : m.i;
}


}  // namespace lpx
