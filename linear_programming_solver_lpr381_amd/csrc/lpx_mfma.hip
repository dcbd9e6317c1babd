// lpx_mfma.hip -- FP64 matrix-core GEMM for the fast refactorisation of the revised path (K7' fast form).
//
// The reference re-inverts the basis matrix from scratch every iteration (Invert, Models/RevisedPrimalSimplex.cs:402-456,
// :128).  The engine keeps B^-1 by rank-1 updates and only has to RESTORE its accuracy now and then; with an inverse that
// is already good to ~1e-9 one Newton-Schulz step
//
//        R = I - B X ,   X' = X + X R            (residual of X' = R^2: quadratic)
//
// does that with two dense m x m x m contractions -- the "dense panel contraction" for which north_star admits MFMA.
// They run on v_mfma_f64_16x16x4_f64: block tile 128 x 128, K-step 16, four waves of 64 x 64 (4 x 4 MFMA tiles, 128
// accumulator VGPRs per lane), operands staged through double-buffered LDS (one barrier per K-step) with the next K-step's
// global loads in flight during the MFMAs.  An FP64 16x16x4 MFMA keeps a SIMD busy for 64 cycles (32 flop / clk / SIMD, the FP64 vector rate: 78.6 TFLOP/s
// on the chip), so LDS and L2 traffic stay far below their limits and the loop is matrix-pipe bound.
// The exact, bit-faithful Gauss-Jordan (inv_select + lpx_update in lpx_revised.hip) stays the parity mode and the
// fallback when R is not small.  Rounding of this path differs from the reference's Invert (FMA accumulation): it is
// held to  pivots equal / z within 1e-9 / |B^-1 B - I| <= 1e-9, not bitwise.
#include "lpx_internal.h"

namespace lpx {

typedef double mf_d4 __attribute__((ext_vector_type(4)));
typedef double mf_d2 __attribute__((ext_vector_type(2)));

static constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 16;
static constexpr int GM_NT = 256;
static constexpr int GM_LDA = GM_BK + 2;          // LDS row strides (doubles): keep 16-byte alignment, spread the banks
static constexpr int GM_LDB = GM_BN + 4;

// C = I - A*B (mode 0, *absmax receives max |C_ij| as the bits of a non-negative double)  or  C = D + A*B (mode 1).
// A: M x K (lda), B: K x N (ldb), row-major; leading dimensions are multiples of 2 and rows are readable up to a
// multiple of 16 columns (padding may hold anything: columns >= K of A are masked, columns >= N of C are not stored).
__global__ __launch_bounds__(GM_NT) void dgemm_mfma_f64(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb,
                                                        double* __restrict__ C, int ldc, const double* __restrict__ D, int ldd,
                                                        int M, int N, int K, int mode, unsigned long long* absmax)
{
    __shared__ __align__(16) double As2[2][GM_BM * GM_LDA];      // double buffered: one barrier per K-step
    __shared__ __align__(16) double Bs2[2][GM_BK * GM_LDB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;                   // 2 x 2 waves, 64 x 64 each
    const int bm = blockIdx.y * GM_BM, bn = blockIdx.x * GM_BN;
    // staging assignment: A tile 128 x 16 -> thread t: row t/2, 8 consecutive k (4 x double2);  B tile 16 x 128 -> row t/16, 8 columns
    const int ar = t >> 1, ak = (t & 1) * 8;
    const int br = t >> 4, bc = (t & 15) * 8;
    const double* ap = A + (size_t)min(bm + ar, M - 1) * lda;
    mf_d2 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + ak + 2 * u;
            mf_d2 v = {0.0, 0.0};
            if (k < K) { v = *reinterpret_cast<const mf_d2*>(ap + k); if (k + 1 >= K) v.y = 0.0; }
            ra[u] = v;
        }
        const int kb = k0 + br;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = bn + bc + 2 * u;
            mf_d2 v = {0.0, 0.0};
            if (kb < K && c < N) { v = *reinterpret_cast<const mf_d2*>(B + (size_t)kb * ldb + c); if (c + 1 >= N) v.y = 0.0; }
            rb[u] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<mf_d2*>(&As2[buf][ar * GM_LDA + ak + 2 * u]) = ra[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<mf_d2*>(&Bs2[buf][br * GM_LDB + bc + 2 * u]) = rb[u];
    };
    mf_d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mf_d4{0.0, 0.0, 0.0, 0.0};

    const int KT = (K + GM_BK - 1) / GM_BK;
    gload(0);
    lstore(0);
    __syncthreads();
    const int lr = lane & 15, lk = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) gload((kt + 1) * GM_BK);               // in flight while the matrix pipe works
        const double* As = As2[kt & 1];
        const double* Bs = Bs2[kt & 1];
#pragma unroll
        for (int kk = 0; kk < GM_BK; kk += 4) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[(wm * 64 + i * 16 + lr) * GM_LDA + kk + lk];     // A[row = lane&15][k = lane>>4]
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[(kk + lk) * GM_LDB + wn * 64 + j * 16 + lr];     // B[k = lane>>4][col = lane&15]
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < KT) lstore((kt + 1) & 1);                  // the other buffer: nobody reads it during this step
        __syncthreads();
    }
    // epilogue: C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
    double mx = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = bm + wm * 64 + i * 16 + lk + 4 * r;
                const int col = bn + wn * 64 + j * 16 + lr;
                if (row < M && col < N) {
                    double v;
                    if (mode == 0) { v = ((row == col) ? 1.0 : 0.0) - acc[i][j][r]; mx = fmax(mx, fabs(v)); }
                    else v = D[(size_t)row * ldd + col] + acc[i][j][r];
                    C[(size_t)row * ldc + col] = v;
                }
            }
    if (mode == 0 && absmax) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d, 64));
        if (lane == 0) atomicMax(absmax, (unsigned long long)__double_as_longlong(mx));
    }
}

hipError_t launch_dgemm_mfma(const double* A, int lda, const double* B, int ldb, double* C, int ldc, const double* D, int ldd,
                             int M, int N, int K, int mode, unsigned long long* absmax, hipStream_t s)
{
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM);
    hipLaunchKernelGGL(dgemm_mfma_f64, grid, dim3(GM_NT), 0, s, A, lda, B, ldb, C, ldc, D, ldd, M, N, K, mode, absmax);
    return hipGetLastError();
}

}  // namespace lpx
