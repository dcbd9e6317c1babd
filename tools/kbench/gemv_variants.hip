// Microbenchmark (diagnostic): y[j] = c[j] - dot(AT[j,:], pi) for an n x m row-major AT (the pricing GEMV of the
// revised path) -- access shapes and cache policies.  ./gemv_variants [n m ld reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <functional>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ d2 ld16(const double* p)
{
    d2 v;
    if (MODE == 0) v = *reinterpret_cast<const d2*>(p);
    if (MODE == 1) v = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p));
    return v;
}
template <int MODE> __device__ __forceinline__ void waitld() {}

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    return x;
}

// V0: wave per row, UN chunks of 1 KiB in flight, pi from global (L2) each time
template <int UN, int LM, int NT>
__global__ __launch_bounds__(NT) void gemv_wave_row(const double* __restrict__ AT, int ld, int n, int m, const double* __restrict__ pi,
                                                    const double* __restrict__ c, double* __restrict__ y)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int j = blockIdx.x * (NT / 64) + wave;
    if (j >= n) return;
    const double* a = AT + (size_t)j * ld;
    double s[UN]; for (int u = 0; u < UN; ++u) s[u] = 0.0;
    for (int k = lane * 2; k < m; k += 128 * UN) {
        d2 x[UN], p[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const int kk = min(k + 128 * u, m - 2); x[u] = ld16<LM>(a + kk); p[u] = *reinterpret_cast<const d2*>(pi + kk); }
        waitld<LM>();
#pragma unroll
        for (int u = 0; u < UN; ++u) if (k + 128 * u < m) { s[u] += x[u].x * p[u].x; s[u] += x[u].y * p[u].y; }
    }
    double t = 0; for (int u = 0; u < UN; ++u) t += s[u];
    t = wave_sum(t);
    if (lane == 0) y[j] = c[j] - t;
}

// V2: persistent block sweeps rows; wave w owns the k-slice [w*m/NW, (w+1)*m/NW) with its pi slice in REGISTERS for all its rows;
// per row each wave has PER = m/(NW*128) chunks of 1 KiB in flight; partial sums meet in LDS.
template <int NW, int PER, int LM, int RPB>
__global__ __launch_bounds__(NW * 64) void gemv_block_row(const double* __restrict__ AT, int ld, int n, int m, const double* __restrict__ pi,
                                                          const double* __restrict__ c, double* __restrict__ y)
{
    __shared__ double part[RPB][NW];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k0 = wave * PER * 128 + lane * 2;
    d2 p[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) p[u] = *reinterpret_cast<const d2*>(pi + k0 + 128 * u);
    for (int j0 = blockIdx.x * RPB; j0 < n; j0 += gridDim.x * RPB) {
        d2 x[RPB][PER];
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
            const double* a = AT + (size_t)min(j0 + r, n - 1) * ld + k0;
#pragma unroll
            for (int u = 0; u < PER; ++u) x[r][u] = ld16<LM>(a + 128 * u);
        }
        waitld<LM>();
#pragma unroll
        for (int r = 0; r < RPB; ++r) {
            double s = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u) { s += x[r][u].x * p[u].x; s += x[r][u].y * p[u].y; }
            s = wave_sum(s);
            if (lane == 0) part[r][wave] = s;
        }
        __syncthreads();
        if (threadIdx.x < RPB && j0 + threadIdx.x < n) {
            double t = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += part[threadIdx.x][w];
            y[j0 + threadIdx.x] = c[j0 + threadIdx.x] - t;
        }
        __syncthreads();
    }
}

// V3: column-tile form (the update kernel's shape): wave = ROWS rows x 128 k, pi slice in registers, per-row partials to a
// [n][m/128] buffer, second kernel sums them in fixed order
template <int ROWS, int LM>
__global__ __launch_bounds__(64) void gemv_tile(const double* __restrict__ AT, int ld, int n, int m, const double* __restrict__ pi,
                                                double* __restrict__ partial, int nkc)
{
    const int lane = threadIdx.x;
    const int unit = blockIdx.x;
    const int kc = unit % nkc, rb = unit / nkc;
    const int k = kc * 128 + lane * 2;
    const d2 p = *reinterpret_cast<const d2*>(pi + k);
    d2 x[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) x[r] = ld16<LM>(AT + (size_t)min(rb * ROWS + r, n - 1) * ld + k);
    waitld<LM>();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        double s = x[r].x * p.x; s += x[r].y * p.y;
        s = wave_sum(s);
        if (lane == 0 && rb * ROWS + r < n) partial[(size_t)(rb * ROWS + r) * nkc + kc] = s;
    }
}
__global__ void gemv_tile_sum(const double* __restrict__ partial, int nkc, int n, const double* __restrict__ c, double* __restrict__ y)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double t = 0; for (int k = 0; k < nkc; ++k) t += partial[(size_t)j * nkc + k];
    y[j] = c[j] - t;
}

// rank-1 update of W (the tableau path's kernel shape: wave = 8 rows x 128 columns), plain policy
__global__ __launch_bounds__(256) void upd_w(double* __restrict__ T, int ld, int R, const double* __restrict__ prow, const double* __restrict__ fac, int ncw, int nrb)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;
    if (unit >= ncw * nrb) return;
    const int cw = unit % ncw, rb = unit / ncw;
    const int col = cw * 128 + lane * 2;
    if (col >= ld) return;
    const d2 p = *reinterpret_cast<const d2*>(prow + col);
    d2 v[8]; double f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int i = min(rb * 8 + k, R - 1); v[k] = *reinterpret_cast<const d2*>(T + (size_t)i * ld + col); f[k] = fac[i]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int i = rb * 8 + k; if (i < R) { d2 o; o.x = v[k].x - f[k] * p.x; o.y = v[k].y - f[k] * p.y; *reinterpret_cast<d2*>(T + (size_t)i * ld + col) = o; } }
}

struct Var { std::string name; std::function<void()> launch; };

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 8192, m = argc > 2 ? atoi(argv[2]) : 4096;
    const int ld = argc > 3 ? atoi(argv[3]) : (m + 15) / 16 * 16, reps = argc > 4 ? atoi(argv[4]) : 40;
    const size_t tot = (size_t)n * ld;
    double *AT, *pi, *c, *y, *partial, *other;
    CK(hipMalloc(&AT, tot * 8)); CK(hipMalloc(&pi, ld * 8)); CK(hipMalloc(&c, n * 8)); CK(hipMalloc(&y, n * 8));
    CK(hipMalloc(&partial, (size_t)n * 64 * 8));
    CK(hipMalloc(&other, 300u << 20));
    std::vector<double> h(tot); for (size_t i = 0; i < tot; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0;
    CK(hipMemcpy(AT, h.data(), tot * 8, hipMemcpyHostToDevice));
    std::vector<double> hp(ld, 0.5), hc(n, 1.0);
    CK(hipMemcpy(pi, hp.data(), ld * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc.data(), n * 8, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    std::vector<Var> vars;
#define ADD_WR(UN, LM, NT) vars.push_back({"wave_row un=" #UN " lm=" #LM " nt=" #NT, [=] { hipLaunchKernelGGL((gemv_wave_row<UN, LM, NT>), dim3((n + NT / 64 - 1) / (NT / 64)), dim3(NT), 0, s, AT, ld, n, m, pi, c, y); }});
    ADD_WR(4, 0, 256) ADD_WR(4, 1, 256) ADD_WR(8, 0, 256) ADD_WR(8, 1, 256) ADD_WR(8, 1, 64) ADD_WR(4, 1, 64) ADD_WR(2, 1, 64) ADD_WR(16, 1, 64) ADD_WR(16, 1, 256)
#define ADD_BR(NW, PER, LM, RPB, GRID) if (NW * PER * 128 == m) vars.push_back({"block_row nw=" #NW " per=" #PER " lm=" #LM " rpb=" #RPB " grid=" #GRID, [=] { hipLaunchKernelGGL((gemv_block_row<NW, PER, LM, RPB>), dim3(GRID), dim3(NW * 64), 0, s, AT, ld, n, m, pi, c, y); }});
    ADD_BR(4, 8, 0, 1, 2048) ADD_BR(4, 8, 1, 1, 2048) ADD_BR(4, 8, 1, 1, 1024) ADD_BR(4, 8, 1, 2, 1024) ADD_BR(8, 4, 1, 1, 2048) ADD_BR(8, 4, 1, 2, 1024) ADD_BR(8, 4, 1, 4, 1024)
    ADD_BR(16, 2, 1, 4, 512) ADD_BR(16, 2, 1, 2, 512) ADD_BR(16, 2, 1, 8, 256) ADD_BR(8, 4, 0, 2, 1024) ADD_BR(16, 2, 1, 4, 256) ADD_BR(8, 4, 1, 2, 2048) ADD_BR(8, 4, 1, 4, 512)
#define ADD_T(ROWS, LM) { const int nkc = m / 128; const int nb = nkc * ((n + ROWS - 1) / ROWS); vars.push_back({"tile rows=" #ROWS " lm=" #LM " (+sum kernel)", [=] { \
        hipLaunchKernelGGL((gemv_tile<ROWS, LM>), dim3(nb), dim3(64), 0, s, AT, ld, n, m, pi, partial, nkc); hipLaunchKernelGGL(gemv_tile_sum, dim3((n + 255) / 256), dim3(256), 0, s, partial, nkc, n, c, y); }}); }
    ADD_T(4, 1) ADD_T(8, 1) ADD_T(3, 1) ADD_T(4, 0)
    // W side: (m+1) x ldw matrix, ftran = wave_row GEMV over its first m rows, then the rank-1 update
    const int ldw = (m + 1 + 15) / 16 * 16;
    double *W, *aq, *d, *prw;
    CK(hipMalloc(&W, (size_t)(m + 1) * ldw * 8)); CK(hipMalloc(&aq, ldw * 8)); CK(hipMalloc(&d, (m + 1) * 8)); CK(hipMalloc(&prw, ldw * 8));
    CK(hipMemset(W, 0, (size_t)(m + 1) * ldw * 8)); CK(hipMemset(aq, 0, ldw * 8)); CK(hipMemset(d, 0, (m + 1) * 8)); CK(hipMemset(prw, 0, ldw * 8));
    std::function<void()> ftran0 = [=] { hipLaunchKernelGGL((gemv_wave_row<4, 0, 256>), dim3((m + 3) / 4), dim3(256), 0, s, W, ldw, m, m, aq, c, d); };
    std::function<void()> ftran1 = [=] { hipLaunchKernelGGL((gemv_block_row<4, 8, 0, 1>), dim3(2048), dim3(256), 0, s, W, ldw, m, m, aq, c, d); };
    auto updw = [=] { const int ncw = (ldw + 127) / 128, nrb = (m + 1 + 7) / 8; hipLaunchKernelGGL(upd_w, dim3((ncw * nrb + 3) / 4), dim3(256), 0, s, W, ldw, m + 1, prw, d, ncw, nrb); };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<hipEvent_t> ev(4 * reps); for (auto& e : ev) CK(hipEventCreate(&e));
    const double bytes = 8.0 * n * m;
    const bool seq = argc > 5 && atoi(argv[5]) == 1;
    if (seq) {
        printf("SEQUENCE price(AT) -> ftran(W) -> update(W), per-kernel times inside the loop; n=%d m=%d ld=%d\n", n, m, ld);
        for (int fv = 0; fv < 2; ++fv)
        for (auto& v : vars) {
            for (int i = 0; i < 3; ++i) { v.launch(); (fv ? ftran1 : ftran0)(); updw(); }
            CK(hipStreamSynchronize(s));
            for (int i = 0; i < reps; ++i) {
                CK(hipEventRecord(ev[4 * i], s)); v.launch(); CK(hipEventRecord(ev[4 * i + 1], s));
                (fv ? ftran1 : ftran0)(); CK(hipEventRecord(ev[4 * i + 2], s)); updw(); CK(hipEventRecord(ev[4 * i + 3], s));
            }
            CK(hipStreamSynchronize(s)); CK(hipGetLastError());
            double t[3] = {0, 0, 0};
            for (int i = 0; i < reps; ++i) for (int k = 0; k < 3; ++k) { float ms; CK(hipEventElapsedTime(&ms, ev[4 * i + k], ev[4 * i + k + 1])); t[k] += ms; }
            printf("ftran=%s  %-50s price %7.2f us (%6.0f GB/s)  ftran %6.2f us  updW %6.2f us  sum %7.2f\n", fv ? "block_row" : "wave_row ", v.name.c_str(),
                   1e3 * t[0] / reps, bytes / (1e3 * t[0] / reps) / 1e3, 1e3 * t[1] / reps, 1e3 * t[2] / reps, 1e3 * (t[0] + t[1] + t[2]) / reps);
            fflush(stdout);
        }
        return 0;
    }
    printf("n=%d m=%d ld=%d  algorithmic bytes %.1f MB\n", n, m, ld, bytes / 1e6);
    for (int pass = 0; pass < 2; ++pass)
        for (auto& v : vars) {
            for (int i = 0; i < 3; ++i) v.launch();
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) { v.launch(); }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / reps;
            printf("pass %d  %-52s %8.2f us  %7.1f GB/s\n", pass, v.name.c_str(), us, bytes / us / 1e3);
            fflush(stdout);
        }
    return 0;
}
