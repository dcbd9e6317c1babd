// Microbenchmark (diagnostic): cache-policy bits on the streaming stores / loads of the rank-1 update on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ void st16(double* p, d2 v)
{
    if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ d2 ld16(const double* p)
{
    d2 v;
    if (MODE == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (MODE == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int SM> __global__ __launch_bounds__(256) void fill(double* a, size_t n2)
{
    d2 v; v.x = 1.0; v.y = 2.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) st16<SM>(a + 2 * i, v);
}

// the update in the shape of the product kernel: wave = ROWS rows x 128 columns
template <int ROWS, int NT, int LM, int SM>
__global__ __launch_bounds__(NT) void upd(double* __restrict__ T, int ld, int R, const double* __restrict__ prow,
                                          const double* __restrict__ fac, int r, int ncw, int nrb)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * (NT / 64) + wave;
    if (unit >= ncw * nrb) return;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const int row0 = rb * ROWS;
    const double2 p = *reinterpret_cast<const double2*>(prow + col);
    d2 v[ROWS]; double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) { const int i = min(row0 + k, R - 1); v[k] = ld16<LM>(T + (size_t)i * ld + col); f[k] = fac[i]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) { d2 o; o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; st16<SM>(T + (size_t)i * ld + col, o); }
    }
}

struct Var { std::string name; std::function<void()> launch; double bytes; };

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 4097, C = argc > 2 ? atoi(argv[2]) : 12289, reps = argc > 3 ? atoi(argv[3]) : 40;
    const int ld = (C + 15) / 16 * 16;
    const size_t n = (size_t)R * ld;
    double *T, *prow, *fac;
    CK(hipMalloc(&T, n * 8)); CK(hipMalloc(&prow, ld * 8)); CK(hipMalloc(&fac, R * 8));
    CK(hipMemset(T, 0, n * 8));
    std::vector<double> hp(ld, 1e-6), hf(R, 1e-6);
    CK(hipMemcpy(prow, hp.data(), ld * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fac, hf.data(), R * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Var> vars;
    const char* nm[8] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0", "sc1 nt", "sc0 nt"};
#define ADD_FILL(SM) vars.push_back({std::string("fill store=") + nm[SM], [=] { hipLaunchKernelGGL(fill<SM>, dim3(2048), dim3(256), 0, s, T, n / 2); }, 8.0 * R * C});
    ADD_FILL(0) ADD_FILL(1) ADD_FILL(2) ADD_FILL(3) ADD_FILL(4) ADD_FILL(5) ADD_FILL(6) ADD_FILL(7)
#define ADD_UPD(ROWS, NT, LM, SM) { const int ncw = (ld + 127) / 128, nrb = (R + ROWS - 1) / ROWS; const int nb = (ncw * nrb + NT / 64 - 1) / (NT / 64); \
    vars.push_back({std::string("upd rows=" #ROWS " nt=" #NT " load=") + nm[LM] + " store=" + nm[SM], [=] { hipLaunchKernelGGL((upd<ROWS, NT, LM, SM>), dim3(nb), dim3(NT), 0, s, T, ld, R, prow, fac, 7, ncw, nrb); }, 16.0 * R * C}); }
    ADD_UPD(8, 256, 0, 0) ADD_UPD(4, 64, 1, 1)
    ADD_UPD(2, 64, 1, 1) ADD_UPD(3, 64, 1, 1) ADD_UPD(5, 64, 1, 1) ADD_UPD(6, 64, 1, 1) ADD_UPD(8, 64, 1, 1) ADD_UPD(1, 64, 1, 1)
    ADD_UPD(4, 128, 1, 1) ADD_UPD(2, 128, 1, 1) ADD_UPD(4, 256, 1, 1) ADD_UPD(2, 256, 1, 1) ADD_UPD(8, 128, 1, 1)
    ADD_UPD(4, 64, 1, 4) ADD_UPD(4, 64, 1, 6) ADD_UPD(4, 64, 6, 6) ADD_UPD(4, 64, 7, 7)
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("R=%d C=%d ld=%d\n", R, C, ld);
    for (int pass = 0; pass < 2; ++pass)
        for (auto& v : vars) {
            for (int i = 0; i < 3; ++i) v.launch();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.launch();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps;
            printf("pass %d  %-56s %8.2f us  %7.1f GB/s\n", pass, v.name.c_str(), us, v.bytes / us / 1e3);
            fflush(stdout);
        }
    return 0;
}
