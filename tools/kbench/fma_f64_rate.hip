// Microbenchmark (diagnostic): issue rate of v_fma_f64 (vector FP64) on gfx950, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__global__ __launch_bounds__(256) void rate(double* out, int iters, double a0, double b0)
{
    double acc[N];
    for (int i = 0; i < N; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const double a = a0, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0; for (int i = 0; i < N; ++i) s += acc[i];
    if (s == 1.2345e-300) out[0] = s;
}
template <int N> void run(int blocksPerCU)
{
    double* out; hipMalloc(&out, 8);
    const int iters = 20000, blocks = 256 * blocksPerCU;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<N>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.999999, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(rate<N>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999999, 1e-9); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 256 * iters * N * 2.0;
    printf("v_fma_f64: %d chains, %d blocks/CU: %.3f ms  %.1f TFLOP/s\n", N, blocksPerCU, ms, flop / ms / 1e9);
}
int main() { run<8>(1); run<16>(1); run<16>(2); run<32>(2); return 0; }
