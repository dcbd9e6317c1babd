// Microbenchmark (diagnostic, not product code): variants of the rank-1 update T[i,:] -= f[i]*p[:] on an R x C f64
// tableau, to find the access shape that streams fastest through HBM on gfx950.  hipcc --offload-arch=gfx950 -O3
// -ffp-contract=off upd_variants.hip -o upd_variants ; ./upd_variants [R C reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// wave = ROWS rows x (128*CPL) columns; lane owns CPL double2 per row (stride 128 doubles apart -> each load instruction 1 KiB contiguous)
template <int ROWS, int CPL, int NT, bool TRANSPOSED>
__global__ __launch_bounds__(NT) void upd(double* __restrict__ T, int ld, int R, const double* __restrict__ prow,
                                          const double* __restrict__ fac, int r, int ncw, int nrb)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * (NT / 64) + wave;
    if (unit >= ncw * nrb) return;
    int cw, rb;
    if (TRANSPOSED) { rb = unit % nrb; cw = unit / nrb; } else { cw = unit % ncw; rb = unit / ncw; }
    const int col0 = cw * 128 * CPL + lane * 2;
    const int row0 = rb * ROWS;
    double2 p[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const int col = col0 + c * 128; p[c] = col < ld ? *reinterpret_cast<const double2*>(prow + col) : make_double2(0, 0); }
    double2 v[ROWS][CPL]; double f[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R) {
            f[k] = fac[i];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const int col = col0 + c * 128; if (col < ld) v[k][c] = *reinterpret_cast<const double2*>(T + (size_t)i * ld + col); }
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int i = row0 + k;
        if (i < R && i != r) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int col = col0 + c * 128;
                if (col < ld) {
                    double2 o; o.x = v[k][c].x - f[k] * p[c].x; o.y = v[k][c].y - f[k] * p[c].y;
                    *reinterpret_cast<double2*>(T + (size_t)i * ld + col) = o;
                }
            }
        }
    }
}

// persistent: grid-stride over units, ROWS x 128 per wave per step, software-pipelined one unit ahead
template <int ROWS, int NT>
__global__ __launch_bounds__(NT) void upd_persist(double* __restrict__ T, int ld, int R, const double* __restrict__ prow,
                                                  const double* __restrict__ fac, int r, int ncw, int nrb)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = gridDim.x * (NT / 64);
    const int nunits = ncw * nrb;
    for (int unit = blockIdx.x * (NT / 64) + wave; unit < nunits; unit += nw) {
        const int cw = unit % ncw, rb = unit / ncw;
        const int col = cw * 128 + lane * 2;
        if (col >= ld) continue;
        const int row0 = rb * ROWS;
        const double2 p = *reinterpret_cast<const double2*>(prow + col);
        double2 v[ROWS]; double f[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { const int i = row0 + k; if (i < R) { v[k] = *reinterpret_cast<const double2*>(T + (size_t)i * ld + col); f[k] = fac[i]; } }
#pragma unroll
        for (int k = 0; k < ROWS; ++k) { const int i = row0 + k; if (i < R && i != r) { double2 o; o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; *reinterpret_cast<double2*>(T + (size_t)i * ld + col) = o; } }
    }
}

// row-contiguous persistent: each workgroup owns a contiguous span of rows; waves sweep whole rows (each wave: one row at a time,
// 1 KiB per instruction, UN instructions in flight)
template <int UN, int NT>
__global__ __launch_bounds__(NT) void upd_rowsweep(double* __restrict__ T, int ld, int R, const double* __restrict__ prow,
                                                   const double* __restrict__ fac, int r)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nw = gridDim.x * (NT / 64);
    const int gw = blockIdx.x * (NT / 64) + wave;
    // flat space of 128-column chunks in row-major order: chunk id = i * ncw + cw ; wave takes UN consecutive chunks per step
    const int ncw = (ld + 127) / 128;
    const long long total = (long long)R * ncw;
    for (long long c0 = (long long)gw * UN; c0 < total; c0 += (long long)nw * UN) {
        double2 v[UN], p[UN]; double f[UN]; int ii[UN]; int cc[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long c = c0 + u;
            const int i = (int)(c / ncw), cw = (int)(c % ncw);
            ii[u] = (c < total) ? i : -1; cc[u] = cw * 128 + lane * 2;
            if (ii[u] >= 0 && cc[u] < ld) { v[u] = *reinterpret_cast<const double2*>(T + (size_t)i * ld + cc[u]); p[u] = *reinterpret_cast<const double2*>(prow + cc[u]); f[u] = fac[i]; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (ii[u] >= 0 && cc[u] < ld && ii[u] != r) { double2 o; o.x = v[u].x - f[u] * p[u].x; o.y = v[u].y - f[u] * p[u].y; *reinterpret_cast<double2*>(T + (size_t)ii[u] * ld + cc[u]) = o; }
        }
    }
}

__global__ void read4(const double2* __restrict__ a, double* out, size_t n)
{
    double acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; acc += v.x + v.y; }
    if (acc == 1.2345e-300) out[0] = acc;
}
__global__ void fill4(double2* __restrict__ a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_double2(1.0, 2.0);
}
__global__ void copy4(const double2* __restrict__ a, double2* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void rmw4(double2* __restrict__ a, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = a[i]; v.x -= 1e-9; v.y -= 1e-9; a[i] = v; }
}

struct Var { std::string name; std::function<void()> launch; };

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 4097, C = argc > 2 ? atoi(argv[2]) : 12289, reps = argc > 3 ? atoi(argv[3]) : 40;
    const int ld = (C + 15) / 16 * 16;
    const size_t n = (size_t)R * ld;
    double *T, *T2, *prow, *fac;
    CK(hipMalloc(&T, n * 8)); CK(hipMalloc(&T2, n * 8)); CK(hipMalloc(&prow, ld * 8)); CK(hipMalloc(&fac, R * 8));
    std::vector<double> h(n); for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(T, h.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<double> hp(ld, 1e-6), hf(R, 1e-6);
    CK(hipMemcpy(prow, hp.data(), ld * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(fac, hf.data(), R * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int r = 7;
    std::vector<Var> vars;
#define ADD_UPD(ROWS, CPL, NT, TR) { const int ncw = (ld + 128 * CPL - 1) / (128 * CPL), nrb = (R + ROWS - 1) / ROWS; const int nb = (ncw * nrb + NT / 64 - 1) / (NT / 64); \
    vars.push_back({"upd rows=" #ROWS " cpl=" #CPL " nt=" #NT " tr=" #TR, [=] { hipLaunchKernelGGL((upd<ROWS, CPL, NT, TR>), dim3(nb), dim3(NT), 0, s, T, ld, R, prow, fac, r, ncw, nrb); }}); }
    ADD_UPD(8, 1, 256, false)
    ADD_UPD(4, 1, 256, false)
    ADD_UPD(2, 1, 256, false)
    ADD_UPD(1, 1, 256, false)
    ADD_UPD(4, 1, 128, false)
    ADD_UPD(4, 1, 64, false)
    ADD_UPD(2, 2, 256, false)
    ADD_UPD(2, 4, 256, false)
    ADD_UPD(2, 4, 128, false)
    ADD_UPD(2, 4, 512, false)
    ADD_UPD(2, 8, 256, false)
    ADD_UPD(1, 4, 256, false)
    ADD_UPD(1, 2, 256, false)
    ADD_UPD(3, 4, 256, false)
    ADD_UPD(2, 3, 256, false)
#define ADD_P(ROWS, NT, MULT) { const int ncw = (ld + 127) / 128, nrb = (R + ROWS - 1) / ROWS; const int nb = 256 * MULT; \
    vars.push_back({"persist rows=" #ROWS " nt=" #NT " blocks=256*" #MULT, [=] { hipLaunchKernelGGL((upd_persist<ROWS, NT>), dim3(nb), dim3(NT), 0, s, T, ld, R, prow, fac, r, ncw, nrb); }}); }
    ADD_P(4, 256, 8)
    ADD_P(2, 256, 8)
#define ADD_RS(UN, NT, MULT) { const int nb = 256 * MULT; \
    vars.push_back({"rowsweep un=" #UN " nt=" #NT " blocks=256*" #MULT, [=] { hipLaunchKernelGGL((upd_rowsweep<UN, NT>), dim3(nb), dim3(NT), 0, s, T, ld, R, prow, fac, r); }}); }
    ADD_RS(4, 256, 8)
    ADD_RS(2, 256, 8)
    vars.push_back({"copy4 T->T2 (read n, write n)", [=] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, s, (const double2*)T, (double2*)T2, n / 2); }});
    vars.push_back({"rmw4 in place", [=] { hipLaunchKernelGGL(rmw4, dim3(256 * 8), dim3(256), 0, s, (double2*)T, n / 2); }});
    vars.push_back({"read4 only (n bytes = half of algorithmic)", [=] { hipLaunchKernelGGL(read4, dim3(256 * 8), dim3(256), 0, s, (const double2*)T, fac, n / 2); }});
    vars.push_back({"fill4 only (n bytes = half of algorithmic)", [=] { hipLaunchKernelGGL(fill4, dim3(256 * 8), dim3(256), 0, s, (double2*)T2, n / 2); }});
    vars.push_back({"hipMemcpyDtoD", [=] { hipMemcpyAsync(T2, T, n * 8, hipMemcpyDeviceToDevice, s); }});

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = 16.0 * R * C;
    printf("R=%d C=%d ld=%d  algorithmic bytes %.1f MB\n", R, C, ld, bytes / 1e6);
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
            printf("pass %d  %-44s %8.2f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", pass, v.name.c_str(), us, bytes / us / 1e3, bytes / us / 1e3 / 8000.0);
            fflush(stdout);
        }
    return 0;
}
