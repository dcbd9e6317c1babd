// Microbenchmark (diagnostic, not product code): does alternating the sweep direction of the rank-1 update from one pivot
// to the next let the 256 MiB Infinity Cache keep the tail of sweep k for the head of sweep k+1?  A cyclic sweep over a
// tableau larger than the cache never hits; a back-and-forth sweep re-touches the most recently used ~cache-size part first.
// Variants: cache policy of loads / stores (default or nontemporal), uniform or by position in the sweep:
//   head  [0, a)      lines expected in the cache (written last by the previous sweep): loads default, stores nt
//   body  [a, 1 - a)  streams through: loads nt, stores nt
//   tail  [1 - a, 1)  to be kept for the next sweep: loads nt, stores default
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sweep_dir.hip -o sweep_dir ; ./sweep_dir [R C reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld2(const double* p)
{ if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2*>(p)); return *reinterpret_cast<const d2*>(p); }
// The nontemporal STORE is written as inline assembly here: hipcc keeps `nt` as metadata on the store and drops it when it
// merges or reorders the stores of an unrolled loop -- the first version of this file measured variants whose code was not
// what their source said (two of three "nt" stores came out with the default policy).  With the instruction spelled out the
// policy of every variant is what its label says (checked in the ISA: llvm-objdump -d, or hipcc -S).  s_nop: the store reads
// its data registers after issue (cdna_hip_programming.md 5.7).  LOADS keep the builtin (an inline-asm load is invisible to
// hipcc's wait counts): the same merging can strip `nt` from loads when ONE kernel mixes load policies, so only variants
// with a uniform load policy are quoted anywhere (exp / thin / cand / the uniform nt and default kernels; their loads were
// checked: three `global_load_dwordx4 ... nt`).  The two `upd` modes and the positional variants that mix LOAD policies
// are exploratory only.
template <bool NT> __device__ __forceinline__ void st2(double* p, d2 v)
{
    if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<d2*>(p) = v;
}

// one wave per block: ROWS rows x 128 columns; mode: 0 = uniform (LNT/SNT), 1 = positional
template <int ROWS, bool LNT, bool SNT>
__device__ __forceinline__ void body(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                     int r, int cw, int rb, int lane)
{
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    d2 v[ROWS]; double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { const int i = rb * ROWS + k; if (i < R) { v[k] = ld2<LNT>(T + (size_t)i * ld + col); f[k] = fac[i]; } }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { const int i = rb * ROWS + k; if (i < R && i != r) { d2 o; o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; st2<SNT>(T + (size_t)i * ld + col, o); } }
}

template <int ROWS>
__global__ __launch_bounds__(64) void upd(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                          int r, int ncw, int nrb, int rev, int mode, int head_units, int tail_units)
{
    const int lane = threadIdx.x;
    const int total = ncw * nrb;
    const int pos = blockIdx.x;                                  // position in the sweep (dispatch order)
    const int unit = rev ? total - 1 - pos : pos;                // row-major tiles: consecutive units are neighbours in memory
    const int cw = unit % ncw, rb = unit / ncw;
    switch (mode) {
    case 0: body<ROWS, true, true>(T, ld, R, prow, fac, r, cw, rb, lane); break;
    case 1: body<ROWS, false, false>(T, ld, R, prow, fac, r, cw, rb, lane); break;
    case 2: body<ROWS, true, false>(T, ld, R, prow, fac, r, cw, rb, lane); break;
    case 3: body<ROWS, false, true>(T, ld, R, prow, fac, r, cw, rb, lane); break;
    default:
        if (pos < head_units) body<ROWS, false, true>(T, ld, R, prow, fac, r, cw, rb, lane);
        else if (pos >= total - tail_units) body<ROWS, true, false>(T, ld, R, prow, fac, r, cw, rb, lane);
        else body<ROWS, true, true>(T, ld, R, prow, fac, r, cw, rb, lane);
    }
}


// the product kernel's prologue in front of the same tile work: state record -> status / pivot row, 128 partials reduced per
// wave to the next entering column (lookahead), ping-pong factor buffer chosen by the iteration parity.  HOIST issues the
// tile loads before any of it (they depend on none of it).
struct St { int status, r, qn_valid, qn, iter, pad[3]; };
template <int ROWS, bool HOIST, bool HOISTP, bool NORED>
__global__ __launch_bounds__(64) void upd_chain(double* __restrict__ T, int ld, int R, const double* __restrict__ prow0, const double* __restrict__ fac0,
                                                const double* __restrict__ fac1, const St* __restrict__ st, const double* __restrict__ part_v,
                                                const int* __restrict__ part_i, double* __restrict__ nxt, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    d2 v[ROWS];
    if (HOIST && col < ld) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { const int i = rb * ROWS + k; if (i < R) v[k] = ld2<true>(T + (size_t)i * ld + col); }
    }
    double xv = 0, yv0 = 0; int xi = 0, yi0 = 0;
    if (HOISTP && !NORED) { xv = part_v[lane]; xi = part_i[lane]; yv0 = part_v[lane + 64]; yi0 = part_i[lane + 64]; }
    const int status = st->status, r = st->r;
    if (status != 4) return;
    int qn;
    if (NORED || st->qn_valid) qn = st->qn;
    else {
        if (!HOISTP) { xv = part_v[lane]; xi = part_i[lane]; yv0 = part_v[lane + 64]; yi0 = part_i[lane + 64]; }
        if (yv0 < xv || (yv0 == xv && yi0 < xi)) { xv = yv0; xi = yi0; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double yv = __shfl_xor(xv, d, 64); const int yi = __shfl_xor(xi, d, 64); if (yv < xv || (yv == xv && yi < xi)) { xv = yv; xi = yi; } }
        qn = xi;
    }
    if (r < 0) return;
    const double* __restrict__ fac = ((st->iter - 1) & 1) ? fac1 : fac0;
    if (col >= ld) return;
    const double* prow = prow0 + (size_t)(r & 0) * ld;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    const bool wq = (qn & ~1) == col;
    double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { const int i = rb * ROWS + k; if (i < R) { if (!HOIST) v[k] = ld2<true>(T + (size_t)i * ld + col); f[k] = fac[i]; } }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = rb * ROWS + k;
        if (i < R) {
            d2 o = p;
            if (i != r) { o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; st2<true>(T + (size_t)i * ld + col, o); }
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
        }
    }
}

// the same prologue flattened: state record (one 32-byte load), both factor buffers, the partials and the tiles are all
// requested before the first branch -- one memory round trip in front of the arithmetic instead of four dependent ones
template <int ROWS, bool HOIST>
__global__ __launch_bounds__(64) void upd_flat(double* __restrict__ T, int ld, int R, const double* __restrict__ prow0, const double* __restrict__ fac0,
                                               const double* __restrict__ fac1, const St* __restrict__ st, const double* __restrict__ part_v,
                                               const int* __restrict__ part_i, double* __restrict__ nxt, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = min(cw * 128 + lane * 2, ld - 2);
    const bool incol = cw * 128 + lane * 2 < ld;
    d2 v[ROWS]; double f0[ROWS], f1[ROWS];
    typedef int i4 __attribute__((ext_vector_type(4)));
    const i4 sa = *reinterpret_cast<const i4*>(st);                 // status, r, qn_valid, qn
    const int iter = st->iter;
    double xv = part_v[lane]; int xi = part_i[lane]; const double yv0 = part_v[lane + 64]; const int yi0 = part_i[lane + 64];
    const d2 p = *reinterpret_cast<const d2*>(prow0 + col);
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = min(rb * ROWS + k, R - 1);
        f0[k] = fac0[i]; f1[k] = fac1[i];
        if (HOIST) v[k] = ld2<true>(T + (size_t)i * ld + col);
    }
    const int status = sa.x, r = sa.y;
    // straight-line: no branch before the stores, so nothing can be sunk behind one
    if (yv0 < xv || (yv0 == xv && yi0 < xi)) { xv = yv0; xi = yi0; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double yv = __shfl_xor(xv, d, 64); const int yi = __shfl_xor(xi, d, 64); if (yv < xv || (yv == xv && yi < xi)) { xv = yv; xi = yi; } }
    const int qn = sa.z ? sa.w : xi;
    const bool live = status == 4 && r >= 0 && incol;
    const bool par = (iter - 1) & 1;
    const bool wq = (qn & ~1) == col;
    if (!HOIST) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { const int i = min(rb * ROWS + k, R - 1); if (live) v[k] = ld2<true>(T + (size_t)i * ld + col); }
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = rb * ROWS + k;
        if (live && i < R) {
            const double f = par ? f1[k] : f0[k];
            d2 o = p;
            if (i != r) { o.x = v[k].x - f * p.x; o.y = v[k].y - f * p.y; st2<true>(T + (size_t)i * ld + col, o); }
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
        }
    }
}

// bisect: which part of the prologue costs throughput?  LEVEL 0 = plain; 1 = + status check from the state record;
// 2 = + pivot row index from the record; 3 = + factor buffer chosen by the record's iteration parity; 4 = + entering column
// from the record and the capture of that column (no reduction)
template <int ROWS, int LEVEL>
__global__ __launch_bounds__(64) void upd_bis(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac0,
                                              const double* __restrict__ fac1, const St* __restrict__ st, double* __restrict__ nxt, int ncw, int nrb, int rarg)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (LEVEL >= 1 && st->status != 4) return;
    const int r = LEVEL >= 2 ? st->r : rarg;
    if (LEVEL >= 2 && r < 0) return;
    const double* __restrict__ fac = (LEVEL >= 3 && ((st->iter - 1) & 1)) ? fac1 : fac0;
    const int qn = LEVEL >= 4 ? st->qn : -2;
    if (col >= ld) return;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    const bool wq = (qn & ~1) == col;
    d2 v[ROWS]; double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { const int i = rb * ROWS + k; if (i < R) { v[k] = ld2<true>(T + (size_t)i * ld + col); f[k] = fac[i]; } }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = rb * ROWS + k;
        if (i < R) {
            d2 o = p;
            if (i != r) { o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; st2<true>(T + (size_t)i * ld + col, o); }
            if (LEVEL >= 4 && wq) nxt[i] = (qn & 1) ? o.y : o.x;
        }
    }
}

// straight-line fast path: a wave whose ROWS rows are all live and do not contain the pivot row runs loads -> arithmetic ->
// stores with no branch in between, so the compiler's wait-count pass keeps exact counts (no "wait for everything" -- which
// also waits for the previous STORE's acknowledgement -- between the stores).  PRO = the shipped prologue in front.
template <int ROWS, bool PRO>
__global__ __launch_bounds__(64) void upd_fast(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac0,
                                               const double* __restrict__ fac1, const St* __restrict__ st, const double* __restrict__ part_v,
                                               const int* __restrict__ part_i, double* __restrict__ nxt, int ncw, int nrb, int rarg)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    int r = rarg, qn = -2;
    const double* __restrict__ fac = fac0;
    if (PRO) {
        const int status = st->status; r = st->r;
        if (status != 4) return;
        if (st->qn_valid) qn = st->qn;
        else {
            double xv = part_v[lane]; int xi = part_i[lane];
            { const double yv = part_v[lane + 64]; const int yi = part_i[lane + 64]; if (yv < xv || (yv == xv && yi < xi)) { xv = yv; xi = yi; } }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { const double yv = __shfl_xor(xv, d, 64); const int yi = __shfl_xor(xi, d, 64); if (yv < xv || (yv == xv && yi < xi)) { xv = yv; xi = yi; } }
            qn = xi;
        }
        if (r < 0) return;
        fac = ((st->iter - 1) & 1) ? fac1 : fac0;
    }
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    const bool wq = (qn & ~1) == col;
    double* base = T + (size_t)row0 * ld + col;
    if (row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS) && !__any(wq)) {
        const d2 p = *reinterpret_cast<const d2*>(prow + col);
        d2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ld2<true>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y; }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) st2<true>(base + (size_t)k * ld, v[k]);
        return;
    }
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            d2 o = p;
            if (i != r) { const d2 v = ld2<true>(base + (size_t)k * ld); const double f = fac[i]; o.x = v.x - f * p.x; o.y = v.y - f * p.y; st2<true>(base + (size_t)k * ld, o); }
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
        }
    }
}

// upd without the mode switch (same arguments, body<true,true> called directly)
template <int ROWS>
__global__ __launch_bounds__(64) void upd_m0(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                             int r, int ncw, int nrb, int rev, int mode, int head_units, int tail_units)
{
    const int lane = threadIdx.x;
    const int total = ncw * nrb;
    const int pos = blockIdx.x;
    const int unit = rev ? total - 1 - pos : pos;
    const int cw = unit % ncw, rb = unit / ncw;
    body<ROWS, true, true>(T, ld, R, prow, fac, r, cw, rb, lane);
}
// upd with a 2-way switch only
template <int ROWS>
__global__ __launch_bounds__(64) void upd_m2(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                             int r, int ncw, int nrb, int rev, int mode, int head_units, int tail_units)
{
    const int lane = threadIdx.x;
    const int total = ncw * nrb;
    const int pos = blockIdx.x;
    const int unit = rev ? total - 1 - pos : pos;
    const int cw = unit % ncw, rb = unit / ncw;
    if (mode == 0) body<ROWS, true, true>(T, ld, R, prow, fac, r, cw, rb, lane);
    else body<ROWS, false, false>(T, ld, R, prow, fac, r, cw, rb, lane);
}

// per-row cache policy masks (bit k set = row k of the wave's three rows uses the nontemporal hint)
template <int ROWS, int LMASK, int SMASK>
__global__ __launch_bounds__(64) void upd_mask(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                               int r, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    double* base = T + (size_t)row0 * ld + col;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    if (row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS)) {
        d2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ((LMASK >> k) & 1) ? ld2<true>(base + (size_t)k * ld) : ld2<false>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y; }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { if ((SMASK >> k) & 1) st2<true>(base + (size_t)k * ld, v[k]); else st2<false>(base + (size_t)k * ld, v[k]); }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) { const d2 v = ld2<true>(base + (size_t)k * ld); const double f = fac[i]; d2 o; o.x = v.x - f * p.x; o.y = v.y - f * p.y; st2<true>(base + (size_t)k * ld, o); }
    }
}

// explicit form: SMASK bit k = store k nontemporal; WMASK bit k = wait for every outstanding memory operation (including
// the previous stores' acknowledgements) before store k
template <int ROWS, int SMASK, int WMASK>
__global__ __launch_bounds__(64) void upd_exp(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                              int r, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    double* base = T + (size_t)row0 * ld + col;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    if (row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS)) {
        d2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ld2<true>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y;
            if ((WMASK >> k) & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((SMASK >> k) & 1) st2<true>(base + (size_t)k * ld, v[k]); else st2<false>(base + (size_t)k * ld, v[k]);
        }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) { const d2 v = ld2<true>(base + (size_t)k * ld); const double f = fac[i]; d2 o; o.x = v.x - f * p.x; o.y = v.y - f * p.y; st2<true>(base + (size_t)k * ld, o); }
    }
}

// product-shaped candidate: shipped prologue (state record, partials reduced to the next entering column, factor parity),
// tile loads optionally issued before it, straight-line fast path with a per-row store policy mask
template <int ROWS, int SMASK, bool HOIST, bool NORED = false, bool HOISTP = false>
__global__ __launch_bounds__(64) void upd_cand(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac0,
                                               const double* __restrict__ fac1, const St* __restrict__ st, const double* __restrict__ part_v,
                                               const int* __restrict__ part_i, double* __restrict__ nxt, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    const int row0 = rb * ROWS;
    const bool inb = col < ld && row0 + ROWS <= R;
    double* base = T + (size_t)row0 * ld + col;
    d2 v[ROWS];
    if (HOIST && inb) {
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ld2<true>(base + (size_t)k * ld);
    }
    double hxv = 0, hyv = 0; int hxi = 0, hyi = 0;
    if (HOISTP && !NORED) { hxv = part_v[lane]; hxi = part_i[lane]; hyv = part_v[lane + 64]; hyi = part_i[lane + 64]; }
    const int status = st->status; const int r = st->r;
    if (status != 4) return;
    int qn;
    if (NORED || st->qn_valid) qn = st->qn;
    else {
        double xv, yv0; int xi, yi0;
        if (HOISTP) { xv = hxv; xi = hxi; yv0 = hyv; yi0 = hyi; } else { xv = part_v[lane]; xi = part_i[lane]; yv0 = part_v[lane + 64]; yi0 = part_i[lane + 64]; }
        if (yv0 < xv || (yv0 == xv && yi0 < xi)) { xv = yv0; xi = yi0; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const double yv = __shfl_xor(xv, d, 64); const int yi = __shfl_xor(xi, d, 64); if (yv < xv || (yv == xv && yi < xi)) { xv = yv; xi = yi; } }
        qn = xi;
    }
    if (r < 0) return;
    const double* __restrict__ fac = ((st->iter - 1) & 1) ? fac1 : fac0;
    if (col >= ld) return;
    const bool wq = (qn & ~1) == col;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    if (inb && (r < row0 || r >= row0 + ROWS) && !__any(wq)) {
        double f[ROWS];
        if (!HOIST) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) v[k] = ld2<true>(base + (size_t)k * ld);
        }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y; }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { if ((SMASK >> k) & 1) st2<true>(base + (size_t)k * ld, v[k]); else st2<false>(base + (size_t)k * ld, v[k]); }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            d2 o = p;
            if (i != r) { const d2 w = ld2<true>(base + (size_t)k * ld); const double f = fac[i]; o.x = w.x - f * p.x; o.y = w.y - f * p.y; st2<true>(base + (size_t)k * ld, o); }
            if (wq) nxt[i] = (qn & 1) ? o.y : o.x;
        }
    }
}

// thinner store mix for tableaux beyond twice the cache: only every MOD-th row block stores its first row with the default
// policy (1 / (3 MOD) of the tableau goes through the Infinity Cache)
template <int ROWS, int MOD>
__global__ __launch_bounds__(64) void upd_thin(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac,
                                               int r, int ncw, int nrb)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    double* base = T + (size_t)row0 * ld + col;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    if (row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS)) {
        d2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ld2<true>(base + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y; }
        if (rb % MOD == 0) st2<false>(base, v[0]); else st2<true>(base, v[0]);
#pragma unroll
        for (int k = 1; k < ROWS; ++k) st2<true>(base + (size_t)k * ld, v[k]);
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) { const d2 v = ld2<true>(base + (size_t)k * ld); const double f = fac[i]; d2 o; o.x = v.x - f * p.x; o.y = v.y - f * p.y; st2<true>(base + (size_t)k * ld, o); }
    }
}

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 4097, C = argc > 2 ? atoi(argv[2]) : 12289, reps = argc > 3 ? atoi(argv[3]) : 60;
    const int ld = (C + 15) / 16 * 16;
    const size_t n = (size_t)R * ld;
    double *T, *prow, *fac;
    CK(hipMalloc(&T, n * 8)); CK(hipMalloc(&prow, ld * 8)); CK(hipMalloc(&fac, R * 8));
    std::vector<double> h(n); for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(T, h.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<double> hp(ld, 1e-6), hf(R, 1e-6);
    CK(hipMemcpy(prow, hp.data(), ld * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fac, hf.data(), R * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    constexpr int ROWS = 3;
    const int ncw = (ld + 127) / 128, nrb = (R + ROWS - 1) / ROWS, total = ncw * nrb;
    const double bytes = 16.0 * R * C, tab_mb = 8.0 * R * ld / 1e6;
    printf("R=%d C=%d ld=%d  tableau %.1f MB, algorithmic bytes per launch %.1f MB, %d units\n", R, C, ld, tab_mb, bytes / 1e6, total);
    struct V { std::string name; int mode; bool alt; double head_mb, tail_mb; };
    std::vector<V> vs = {
        {"nt/nt one direction", 0, false, 0, 0},
        {"nt/nt alternating", 0, true, 0, 0},
        {"default/default one direction", 1, false, 0, 0},
        {"default/default alternating", 1, true, 0, 0},
        {"nt loads, default stores, alternating", 2, true, 0, 0},
        {"default loads, nt stores, alternating", 3, true, 0, 0},
    };
    for (double mb : {64.0, 96.0, 128.0, 160.0, 192.0, 224.0}) vs.push_back({"positional head=tail=" + std::to_string((int)mb) + " MB alternating", 4, true, mb, mb});
    for (double mb : {128.0, 192.0}) vs.push_back({"positional head=tail=" + std::to_string((int)mb) + " MB ONE direction (control)", 4, false, mb, mb});
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass)
        for (auto& v : vs) {
            const int hu = (int)(v.head_mb / tab_mb * total), tu = (int)(v.tail_mb / tab_mb * total);
            int it = 0;
            auto launch = [&] { hipLaunchKernelGGL(upd<ROWS>, dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb, (v.alt && (it & 1)) ? 1 : 0, v.mode, hu, tu); ++it; };
            for (int i = 0; i < 4; ++i) launch();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps;
            printf("pass %d  %-62s %8.2f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", pass, v.name.c_str(), us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
            fflush(stdout);
        }
    {   // product-like prologue
        St hs{4, 7, 0, 0, 5, {0, 0, 0}}; St* dst; CK(hipMalloc(&dst, sizeof(St))); CK(hipMemcpy(dst, &hs, sizeof(St), hipMemcpyHostToDevice));
        double *pv, *nxt, *fac1; int* pi; CK(hipMalloc(&pv, 128 * 8)); CK(hipMalloc(&pi, 128 * 4)); CK(hipMalloc(&nxt, R * 8)); CK(hipMalloc(&fac1, R * 8));
        std::vector<double> hv(128); std::vector<int> hi(128); for (int i = 0; i < 128; ++i) { hv[i] = -1.0 - (i % 7); hi[i] = 100 + i; }
        CK(hipMemcpy(pv, hv.data(), 128 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(pi, hi.data(), 128 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(fac1, hf.data(), R * 8, hipMemcpyHostToDevice));
        struct CV { const char* name; std::function<void()> launch; };
#define CHAIN(H, HP, NR) [=] { hipLaunchKernelGGL((upd_chain<ROWS, H, HP, NR>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }
        std::vector<CV> cv = {
            {"prologue as shipped: state -> partials -> tiles", CHAIN(false, false, false)},
            {"tile loads hoisted above the prologue", CHAIN(true, false, false)},
            {"partial loads issued with the state loads", CHAIN(false, true, false)},
            {"partials with the state, tiles hoisted", CHAIN(true, true, false)},
            {"entering column precomputed (no reduction)", CHAIN(false, false, true)},
            {"entering column precomputed, tiles hoisted", CHAIN(true, false, true)},
            {"flat prologue (all loads before the first branch)", [=] { hipLaunchKernelGGL((upd_flat<ROWS, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }},
            {"flat prologue, tiles hoisted too", [=] { hipLaunchKernelGGL((upd_flat<ROWS, true>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }},
        };
#define BIS(L) {"bisect level " #L, [=] { hipLaunchKernelGGL((upd_bis<ROWS, L>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, nxt, ncw, nrb, 7); }}
        cv.push_back({"plain upd kernel (mode 0) inside this loop", [=] { hipLaunchKernelGGL(upd<ROWS>, dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb, 0, 0, 0, 0); }});
        cv.push_back(BIS(0)); cv.push_back(BIS(1)); cv.push_back(BIS(2)); cv.push_back(BIS(3)); cv.push_back(BIS(4));
        cv.push_back({"straight-line fast path, no prologue", [=] { hipLaunchKernelGGL((upd_fast<ROWS, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb, 7); }});
        cv.push_back({"straight-line fast path behind the shipped prologue", [=] { hipLaunchKernelGGL((upd_fast<ROWS, true>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb, 7); }});
        cv.push_back({"upd without the switch", [=] { hipLaunchKernelGGL(upd_m0<ROWS>, dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb, 0, 0, 0, 0); }});
        cv.push_back({"upd with a two-way switch", [=] { hipLaunchKernelGGL(upd_m2<ROWS>, dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb, 0, 0, 0, 0); }});
        cv.push_back({"exp: stores nt,nt,default; wait before 2nd and 3rd (mimics the fast kernel)", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 3, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,nt,nt; wait before 2nd and 3rd", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 7, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default x3; wait before 2nd and 3rd", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 0, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,nt,default; no waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 3, 0>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,nt,default; wait before 3rd only", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 3, 4>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,nt,nt; wait before 3rd only", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 7, 4>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default,nt,nt; waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 6, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,default,nt; waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 5, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,default,default; waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 1, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default,nt,nt; no waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 6, 0>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores nt,default,nt; no waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 5, 0>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default,nt,nt; wait before 3rd only", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 6, 4>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default,nt,nt; wait before 2nd only", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 6, 2>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"exp: stores default,default,nt; waits", [=] { hipLaunchKernelGGL((upd_exp<ROWS, 4, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"cand: prologue, stores nt x3", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 7, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: prologue, stores nt x3, tiles hoisted", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 7, true>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: prologue, stores default,nt,nt", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 6, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: prologue, stores default,nt,nt, tiles hoisted", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 6, true>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: NO reduction (qn in the record), stores default,nt,nt", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 6, false, true, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: NO reduction, default,nt,nt, tiles hoisted", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 6, true, true, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: reduction, partial + tile loads hoisted, default,nt,nt", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 6, true, false, true>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"cand: NO reduction, stores nt x3", [=] { hipLaunchKernelGGL((upd_cand<ROWS, 7, false, true, false>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, fac1, dst, pv, pi, nxt, ncw, nrb); }});
        cv.push_back({"thin mix: first row default in every 1-th row block (1/3 of the tableau)", [=] { hipLaunchKernelGGL((upd_thin<ROWS, 1>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"thin mix: first row default in every 2-th row block (1/6 of the tableau)", [=] { hipLaunchKernelGGL((upd_thin<ROWS, 2>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"thin mix: first row default in every 3-th row block (1/9 of the tableau)", [=] { hipLaunchKernelGGL((upd_thin<ROWS, 3>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"thin mix: first row default in every 4-th row block (1/12 of the tableau)", [=] { hipLaunchKernelGGL((upd_thin<ROWS, 4>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"thin mix: first row default in every 6-th row block (1/18 of the tableau)", [=] { hipLaunchKernelGGL((upd_thin<ROWS, 6>), dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }});
        cv.push_back({"plain upd kernel (mode 0) again", [=] { hipLaunchKernelGGL(upd<ROWS>, dim3(total), dim3(64), 0, s, T, ld, R, prow, fac, 7, ncw, nrb, 0, 0, 0, 0); }});
        for (int pass = 0; pass < 2; ++pass)
            for (auto& c : cv) {
                for (int i = 0; i < 4; ++i) c.launch();
                CK(hipStreamSynchronize(s));
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < reps; ++i) c.launch();
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                CK(hipGetLastError());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = 1e3 * ms / reps;
                printf("pass %d  %-62s %8.2f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", pass, c.name, us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
                fflush(stdout);
            }
    }
    return 0;
}
