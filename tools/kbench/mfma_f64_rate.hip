// Microbenchmark (diagnostic): issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- back-to-back, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e-300) out[0] = s;
}
template <int NACC> void run(int wavesPerSimd, const char* name)
{
    double* out; hipMalloc(&out, 8);
    const int iters = 20000, blocks = 256 * wavesPerSimd;   // 256 threads = 4 waves = one per SIMD; `wavesPerSimd` blocks per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(rate<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1.0); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * iters * NACC;      // MFMAs issued
    const double flop = mf * 2048.0;
    printf("%s: %d acc, %d waves/SIMD: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, NACC, wavesPerSimd, ms,
           flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (mf / 1024.0));
}
int main() { run<1>(1, "dep-chain"); run<4>(1, "4 acc"); run<8>(1, "8 acc"); run<16>(1, "16 acc"); run<4>(2, "4 acc"); run<16>(2, "16 acc"); return 0; }
