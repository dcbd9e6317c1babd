// Microbenchmark (diagnostic, not product code): could the streaming rank-1 update run OUT OF PLACE (T_k -> T_{k+1} in a
// second buffer, ping-pong), so that the select of pivot k+1 -- which then depends only on T_k and the records of pivot k --
// runs inside the same launch instead of after it?  Questions: (1) is an out-of-place sweep as fast as the in-place one
// under the same load / store policies (the Infinity Cache now sees two 403 MB buffers); (2) what do ~100 select-like
// single-wave workgroups at the head of the grid (strided column gather, divisions, a row slice) cost the sweep.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off oop.hip -o oop ; ./oop [R C reps]
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
// nontemporal stores spelled out (hipcc drops the hint when it merges stores, see sweep_dir.hip)
template <bool NT> __device__ __forceinline__ void st2(double* p, d2 v)
{
    if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else *reinterpret_cast<d2*>(p) = v;
}

// select-like work of one single-wave workgroup: gather a column of `src` (stride ld), one division per element, wave
// minimum; then a 96-column slice of two rows.  `sink` keeps the result alive.
__device__ __forceinline__ void select_like(const double* __restrict__ src, int ld, int R, int C, int q, const double* __restrict__ fac,
                                            const double* __restrict__ prow, int unit, int nsel, double* sink)
{
    const int lane = threadIdx.x;
    double best = 1e300;
    const double pq = prow[q];
    for (int i0 = 0; i0 < R; i0 += 64 * 8) {
        double c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int i = min(R - 1, i0 + k * 64 + lane); c[k] = src[(size_t)i * ld + q] - fac[i] * pq; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const double ratio = (c[k] > 1e-9) ? 1.0 / c[k] : 1e300; best = fmin(best, ratio); }
    }
    for (int o = 32; o; o >>= 1) best = fmin(best, __shfl_xor(best, o));
    const int r = ((int)(best * 1e3) & 0x7fffffff) % R;
    const int per = (C + nsel - 1) / nsel;
    double acc = 0.0;
    for (int j = unit * per + lane; j < min(C, (unit + 1) * per); j += 64) {
        const double p = (src[(size_t)r * ld + j] - fac[r] * prow[j]) / (best + 2.0);
        acc += src[(size_t)(R - 1) * ld + j] - fac[R - 1] * prow[j] - 0.5 * p;
    }
    if (acc == 12345.678) sink[unit] = acc;
}

// The same in the shape a real fused kernel would have: 256-lane workgroups; every wave gathers 1024 rows of the column in ONE
// batch of loads (16 per lane), the waves meet in LDS, the row slice follows, then the last-workgroup hand-off of
// lpx_select_mb (agent-scope partial stores, wait, barrier, one add per workgroup, the last one reads the partials).
// Dependent round trips: prow[q] -> column gather -> row slice -> partial stores -> add -> partial loads.
__device__ __forceinline__ void select_like256(const double* __restrict__ src, int ld, int R, int C, int q, const double* __restrict__ fac,
                                               const double* __restrict__ prow, int unit, int nsel, double* sink, int* cnt)
{
    __shared__ double s_best[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (C + nsel - 1) / nsel;
    const int j = min(C - 1, unit * per + (int)threadIdx.x), j2 = min(C - 1, j + 256);
    // independent of the ratio test: issued first
    const double pj = prow[j], pj2 = prow[j2], oj = src[(size_t)(R - 1) * ld + j], oj2 = src[(size_t)(R - 1) * ld + j2], fm = fac[R - 1];
    const double pq = prow[q];
    double c[16], f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int i = min(R - 1, wave * 1024 + k * 64 + lane); c[k] = src[(size_t)i * ld + q]; f[k] = fac[i]; }
    double best = 1e300;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const double v = c[k] - f[k] * pq; const double ratio = (v > 1e-9) ? 1.0 / v : 1e300; best = fmin(best, ratio); }
    for (int o = 32; o; o >>= 1) best = fmin(best, __shfl_xor(best, o));
    if (lane == 0) s_best[wave] = best;
    __syncthreads();
    best = fmin(fmin(s_best[0], s_best[1]), fmin(s_best[2], s_best[3]));
    const int r = ((int)(best * 1e3) & 0x7fffffff) % R;
    const double fr = fac[r];
    const double p = (src[(size_t)r * ld + j] - fr * pj) / (best + 2.0), p2 = (src[(size_t)r * ld + j2] - fr * pj2) / (best + 2.0);
    double acc = fmin(oj - fm * pj - 0.5 * p, oj2 - fm * pj2 - 0.5 * p2);
    for (int o = 32; o; o >>= 1) acc = fmin(acc, __shfl_xor(acc, o));
    if (lane == 0) __hip_atomic_store(&sink[unit * 4 + wave], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        int last = 0;
        if (threadIdx.x == 0) last = (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsel - 1) ? 1 : 0;
        last = __builtin_amdgcn_readfirstlane(last);
        if (last) {
            double x = __hip_atomic_load(&sink[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane + 64 < nsel * 4) x = fmin(x, __hip_atomic_load(&sink[lane + 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            for (int o = 32; o; o >>= 1) x = fmin(x, __shfl_xor(x, o));
            if (lane == 0) { sink[2048] = x; __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
}

// SMASK bit k = store k nontemporal; OOP: read src, write dst; nsel select-like workgroups first
template <int ROWS, int SMASK, bool LNT, int NTH = 64>
__global__ __launch_bounds__(NTH) void upd(const double* __restrict__ src, double* __restrict__ dst, int ld, int R, int C,
                                          const double* __restrict__ prow, const double* __restrict__ fac,
                                          int r, int ncw, int nsel, int q, double* sink)
{
    if ((int)blockIdx.x < nsel) {
        if (NTH >= 256) select_like256(src, ld, R, C, q, fac, prow, blockIdx.x, nsel, sink, reinterpret_cast<int*>(sink + 3000));
        else select_like(src, ld, R, C, q, fac, prow, blockIdx.x, nsel, sink);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int unit = (blockIdx.x - nsel) * (NTH / 64) + (threadIdx.x >> 6);
    if (unit >= ncw * ((R + ROWS - 1) / ROWS)) return;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    const double* sb = src + (size_t)row0 * ld + col;
    double* db = dst + (size_t)row0 * ld + col;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    if (row0 + ROWS <= R && (r < row0 || r >= row0 + ROWS)) {
        d2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) v[k] = ld2<LNT>(sb + (size_t)k * ld);
#pragma unroll
        for (int k = 0; k < ROWS; ++k) f[k] = fac[row0 + k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            v[k].x = v[k].x - f[k] * p.x; v[k].y = v[k].y - f[k] * p.y;
            if ((SMASK >> k) & 1) st2<true>(db + (size_t)k * ld, v[k]); else st2<false>(db + (size_t)k * ld, v[k]);
        }
        return;
    }
#pragma unroll 1
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            d2 o = p;
            if (i != r) { const d2 v = ld2<true>(sb + (size_t)k * ld); const double f = fac[i]; o.x = v.x - f * p.x; o.y = v.y - f * p.y; }
            st2<true>(db + (size_t)k * ld, o);
        }
    }
}

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 4097, C = argc > 2 ? atoi(argv[2]) : 12289, reps = argc > 3 ? atoi(argv[3]) : 60;
    const int ld = (C + 15) / 16 * 16;
    const size_t n = (size_t)R * ld;
    double *T, *T2, *prow, *fac, *sink;
    CK(hipMalloc(&T, n * 8)); CK(hipMalloc(&T2, n * 8)); CK(hipMalloc(&prow, ld * 8)); CK(hipMalloc(&fac, R * 8)); CK(hipMalloc(&sink, 4096 * 8)); CK(hipMemset(sink, 0, 4096 * 8));
    std::vector<double> h(n); for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(T, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(T2, h.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<double> hp(ld, 1e-6), hf(R, 1e-6);
    CK(hipMemcpy(prow, hp.data(), ld * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fac, hf.data(), R * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    constexpr int ROWS = 3;
    const int ncw = (ld + 127) / 128, nrb = (R + ROWS - 1) / ROWS, total = ncw * nrb;
    const double bytes = 16.0 * R * C;
    printf("R=%d C=%d ld=%d  tableau %.1f MB, algorithmic bytes per launch %.1f MB, %d units\n", R, C, ld, 8.0 * R * ld / 1e6, bytes / 1e6, total);
    struct CV { std::string name; std::function<void(int)> launch; };
    std::vector<CV> cv;
#define VAR(NAME, SM, LNT, OOP, NSEL) cv.push_back({NAME, [=](int it) { \
        const double* a = (OOP && (it & 1)) ? T2 : T; double* b = OOP ? ((it & 1) ? T : T2) : T; \
        hipLaunchKernelGGL((upd<ROWS, SM, LNT>), dim3(total + NSEL), dim3(64), 0, s, a, b, ld, R, C, prow, fac, 7, ncw, NSEL, 1234 + it % 7, sink); }})
    VAR("in place   loads nt, stores nt,nt,default (the shipped mix)", 3, true, false, 0);
    VAR("ping-pong  loads nt, stores nt,nt,default", 3, true, true, 0);
    VAR("ping-pong  loads nt, stores nt,nt,nt", 7, true, true, 0);
    VAR("ping-pong  loads nt, stores nt,default,default", 1, true, true, 0);
    VAR("ping-pong  loads nt, stores default x3", 0, true, true, 0);
    VAR("ping-pong  loads default, stores nt,nt,default", 3, false, true, 0);
    VAR("ping-pong  loads default, stores nt,nt,nt", 7, false, true, 0);
    VAR("ping-pong  loads default, stores default x3", 0, false, true, 0);
    VAR("in place   shipped mix + 128 select-like workgroups", 3, true, false, 128);
    VAR("ping-pong  shipped mix + 128 select-like workgroups", 3, true, true, 128);
    VAR("ping-pong  shipped mix + 256 select-like workgroups", 3, true, true, 256);
    VAR("ping-pong  shipped mix +  32 select-like workgroups", 3, true, true, 32);
#define VAR256(NAME, OOP, NSEL) cv.push_back({NAME, [=](int it) { \
        const double* a = (OOP && (it & 1)) ? T2 : T; double* b = OOP ? ((it & 1) ? T : T2) : T; \
        hipLaunchKernelGGL((upd<ROWS, 3, true, 256>), dim3((total + 3) / 4 + NSEL), dim3(256), 0, s, a, b, ld, R, C, prow, fac, 7, ncw, NSEL, 1234 + it % 7, sink); }})
    VAR256("in place   shipped mix, 256-lane workgroups", false, 0);
    VAR256("ping-pong  shipped mix, 256-lane workgroups", true, 0);
    VAR256("ping-pong  shipped mix, 256-lane workgroups + 32 select-like (real chain)", true, 32);
    VAR256("ping-pong  shipped mix, 256-lane workgroups + 16 select-like (real chain)", true, 16);
#define VARN(NAME, N) cv.push_back({NAME, [=](int it) { \
        const double* a = (it & 1) ? T2 : T; double* b = (it & 1) ? T : T2; \
        hipLaunchKernelGGL((upd<ROWS, 3, true, N>), dim3((total + N / 64 - 1) / (N / 64)), dim3(N), 0, s, a, b, ld, R, C, prow, fac, 7, ncw, 0, 1234, sink); }})
    VARN("ping-pong  shipped mix, 128-lane workgroups", 128);
    VARN("ping-pong  shipped mix, 192-lane workgroups", 192);
    VARN("ping-pong  shipped mix, 512-lane workgroups", 512);
    VAR("in place   shipped mix again", 3, true, false, 0);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass)
        for (auto& c : cv) {
            int it = 0;
            for (int i = 0; i < 6; ++i) c.launch(it++);
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) c.launch(it++);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps;
            printf("pass %d  %-62s %8.2f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", pass, c.name.c_str(), us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
            fflush(stdout);
        }
    CK(hipMemset(sink, 0, 4096 * 8));
    for (int nsel : {16, 32}) {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((upd<ROWS, 3, true, 256>), dim3(nsel), dim3(256), 0, s, T, T2, ld, R, C, prow, fac, 7, ncw, nsel, 1234 + i, sink);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((upd<ROWS, 3, true, 256>), dim3(nsel), dim3(256), 0, s, T, T2, ld, R, C, prow, fac, 7, ncw, nsel, 1234 + i % 7, sink);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("256-lane select-like workgroups (real chain) alone, %3d of them: %8.2f us per launch\n", nsel, 1e3 * ms / reps);
    }
    // the select-like workgroups alone
    for (int nsel : {32, 128, 256}) {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((upd<ROWS, 3, true>), dim3(nsel), dim3(64), 0, s, T, T2, ld, R, C, prow, fac, 7, ncw, nsel, 1234 + i, sink);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((upd<ROWS, 3, true>), dim3(nsel), dim3(64), 0, s, T, T2, ld, R, C, prow, fac, 7, ncw, nsel, 1234 + i % 7, sink);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("select-like workgroups alone, %3d of them: %8.2f us per launch\n", nsel, 1e3 * ms / reps);
    }
    return 0;
}
