// lpx_resident.h -- primitives shared by the resident kernels (lpx_resident.hip, lpx_resident_group.hip):
// tagged-granule exchange through global memory and the 4-wave reductions of 1024-lane workgroups.
#pragma once
#include "lpx_block.h"

namespace lpx {

static constexpr int RS_NT = 1024;
static constexpr unsigned RS_SPIN_MAX = 1u << 21;
static constexpr int RS_FETCH = 4;              // granule pairs in flight per lane while gathering

typedef unsigned long long u64;

typedef unsigned rs_u4 __attribute__((ext_vector_type(4)));

// One double = one 16-byte granule pair {lo32, tag, hi32, tag}: a single write-through dwordx4 store, a single
// dwordx4 sc1 load.  Each 8-byte half validates itself, so it does not matter whether the fabric keeps the 16
// bytes together.
__device__ __forceinline__ void rs_publish(u64* g, double v, unsigned gen)
{
    const u64 bits = (u64)__double_as_longlong(v);
    rs_u4 w;
    w.x = (unsigned)bits; w.y = gen; w.z = (unsigned)(bits >> 32); w.w = gen;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(g), "v"(w) : "memory");
}

// Gathers count (<= RS_FETCH) granule pairs g[idx[u]] of generation `gen`; all loads of a round are in flight
// together.  Returns false when the wait expired.
__device__ __forceinline__ bool rs_gather(const u64* g, const int* idx, int count, unsigned gen, double* out,
                                          unsigned max_spin = RS_SPIN_MAX, unsigned* pend = nullptr)
{
    static_assert(RS_FETCH == 4, "the load group below is written for four granule pairs");
    unsigned pending = pend ? *pend : (1u << count) - 1u;
    const u64* p0 = g + 2 * (size_t)idx[0];
    const u64* p1 = g + 2 * (size_t)idx[count > 1 ? 1 : 0];
    const u64* p2 = g + 2 * (size_t)idx[count > 2 ? 2 : 0];
    const u64* p3 = g + 2 * (size_t)idx[count > 3 ? 3 : 0];
    for (unsigned spin = 0; spin < max_spin && pending; ++spin) {
        rs_u4 w[RS_FETCH];
        asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                     "global_load_dwordx4 %1, %5, off sc1\n\t"
                     "global_load_dwordx4 %2, %6, off sc1\n\t"
                     "global_load_dwordx4 %3, %7, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                     : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                     : "memory");
#pragma unroll
        for (int u = 0; u < RS_FETCH; ++u) {
            if (((pending >> u) & 1u) && w[u].y == gen && w[u].w == gen) {
                out[u] = __longlong_as_double((long long)(((u64)w[u].z << 32) | (u64)w[u].x));
                pending &= ~(1u << u);
            }
        }
        if (pending && spin > 32) __builtin_amdgcn_s_sleep(1);
    }
    if (pend) *pend = pending;
    return pending == 0;
}

// Two granule pairs per round instead of four: half the registers (the register-resident kernel, whose workgroups are as wide as
// the vectors they gather, never needs more than two per lane).
__device__ __forceinline__ bool rs_gather2(const u64* g, int i0, int i1, int count, unsigned gen, double* out,
                                           unsigned max_spin = RS_SPIN_MAX, unsigned* pend = nullptr)
{
    unsigned pending = pend ? *pend : (1u << count) - 1u;
    const u64* p0 = g + 2 * (size_t)i0;
    const u64* p1 = g + 2 * (size_t)(count > 1 ? i1 : i0);
    for (unsigned spin = 0; spin < max_spin && pending; ++spin) {
        rs_u4 w0, w1;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                     "global_load_dwordx4 %1, %3, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(w0), "=&v"(w1) : "v"(p0), "v"(p1) : "memory");
        if ((pending & 1u) && w0.y == gen && w0.w == gen) { out[0] = __longlong_as_double((long long)(((u64)w0.z << 32) | (u64)w0.x)); pending &= ~1u; }
        if ((pending & 2u) && w1.y == gen && w1.w == gen) { out[1] = __longlong_as_double((long long)(((u64)w1.z << 32) | (u64)w1.x)); pending &= ~2u; }
        if (pending && spin > 32) __builtin_amdgcn_s_sleep(1);
    }
    if (pend) *pend = pending;
    return pending == 0;
}

// Three per round: 512 lanes cover a 1536-entry row in one round trip (the three-column form of the register-resident kernel).
__device__ __forceinline__ bool rs_gather3(const u64* g, int i0, int i1, int i2, int count, unsigned gen, double* out,
                                           unsigned max_spin = RS_SPIN_MAX, unsigned* pend = nullptr)
{
    unsigned pending = pend ? *pend : (1u << count) - 1u;
    const u64* p0 = g + 2 * (size_t)i0;
    const u64* p1 = g + 2 * (size_t)(count > 1 ? i1 : i0);
    const u64* p2 = g + 2 * (size_t)(count > 2 ? i2 : i0);
    for (unsigned spin = 0; spin < max_spin && pending; ++spin) {
        rs_u4 w0, w1, w2;
        asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                     "global_load_dwordx4 %1, %4, off sc1\n\t"
                     "global_load_dwordx4 %2, %5, off sc1\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
        if ((pending & 1u) && w0.y == gen && w0.w == gen) { out[0] = __longlong_as_double((long long)(((u64)w0.z << 32) | (u64)w0.x)); pending &= ~1u; }
        if ((pending & 2u) && w1.y == gen && w1.w == gen) { out[1] = __longlong_as_double((long long)(((u64)w1.z << 32) | (u64)w1.x)); pending &= ~2u; }
        if ((pending & 4u) && w2.y == gen && w2.w == gen) { out[2] = __longlong_as_double((long long)(((u64)w2.z << 32) | (u64)w2.x)); pending &= ~4u; }
        if (pending && spin > 32) __builtin_amdgcn_s_sleep(1);
    }
    if (pend) *pend = pending;
    return pending == 0;
}

// Waits until one granule pair carries generation `gen` (same address in every lane: one request per wave).
__device__ __forceinline__ bool rs_wait(const u64* g, unsigned gen)
{
    for (unsigned spin = 0; spin < RS_SPIN_MAX; ++spin) {
        rs_u4 w;
        asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(g) : "memory");
        if (w.y == gen && w.w == gen) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it waits for
// the acknowledgements of the write-through (sc1) granule stores a workgroup has just published -- 1.5-2 us that nobody in the
// workgroup needs: the granules are consumed by OTHER workgroups, which validate them by their tags.
__device__ __forceinline__ void rs_barrier_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Abort flag of a resident launch (DevState::pad[1]), read past the L1 (sc1) like every other cross-workgroup word.
__device__ __forceinline__ int rs_abort_raised(const DevState* st)
{
    int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(&st->pad[1]) : "memory");
    return v;
}
// Diagnostic (LPX_RESIDENT_TEST_MUTE=3): a workgroup that gets its CU only after the others have given up -- it waits
// for the abort flag and then runs as if nothing had happened.
__device__ __forceinline__ void rs_wait_for_abort(const DevState* st)
{
    for (unsigned spin = 0; spin < 8u * RS_SPIN_MAX; ++spin) {
        if (rs_abort_raised(st)) return;
        __builtin_amdgcn_s_sleep(8);
    }
}

// Block reductions of 1024-lane workgroups are slow when all 16 waves take part (four waves per SIMD take turns
// through the same DPP chain, then all of them reduce the partials again: ~1 us).  The small vectors of this
// kernel (objective row, ratios) are reduced by waves 0-3 only -- one per SIMD -- and the other waves just wait.
static constexpr int RS_RT = 256;               // lanes that hold candidates
__device__ __forceinline__ MinIdx first4_min_idx(MinIdx x, double* s_v, int* s_i)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        x = wave_min_idx(x);
        if (lane == 0) { s_v[wave] = x.v; s_i[wave] = x.i; }
    }
    __syncthreads();
    MinIdx y; y.v = s_v[0]; y.i = s_i[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) { MinIdx z; z.v = s_v[k]; z.i = s_i[k]; y = mi_pick(y, z); }
    return y;
}

struct LdsRatio {
    const double* v;
    __device__ __forceinline__ double den(int i) const { return v[i]; }
    __device__ __forceinline__ double num(int) const { return 0.0; }
    __device__ __forceinline__ double value(double a, double) const { return a; }
};

// The exact wave-0 replays are real calls: inlined, their 16-ratio register blocks push the 1024-lane kernels (128
// VGPRs per lane) into scratch spills on the hot path; as calls they cost a few saves only when a tie needs them.
__device__ __attribute__((noinline)) int rs_exact_row_scan(int m, double tol, const double* ratios)
{
    return wave_hysteresis_argmin<LdsRatio, true>(m, tol, LdsRatio{ratios});
}
__device__ __attribute__((noinline)) int rs_exact_col_scan(int L, double tol, const double* row, const double* zrow, double eps)
{
    return wave_hysteresis_argmin<DualColRatio, true>(L, tol, DualColRatio{row, zrow, eps});
}

// The hysteresis scan of Models/PrimalSimplex.cs:234-241 over `ratios[0..m)` in LDS (+inf = ineligible), identical
// in every workgroup.  Fast path (lpx_block.h, wave_hysteresis_argmin): with rmin the smallest ratio and i* its first
// row, if no OTHER row j has fl(r_j - tol) <= rmin the sequential scan ends at i* whatever it accepted on the way.
// Ties and near-ties (the degenerate vertices of 0/1 programs) take the exact scan on wave 0.  All lanes call this.
__device__ __forceinline__ int rs_hysteresis(int m, double tol, const double* ratios, double* s_v, int* s_i, int* s_out)
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
                const int win = rs_exact_row_scan(m, tol, ratios);
                if (t == 0) *s_out = win;
            }
            __syncthreads();
            r = *s_out;
        }
        __syncthreads();
    }
    return r;
}

}  // namespace lpx
