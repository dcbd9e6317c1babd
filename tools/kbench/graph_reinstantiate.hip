// graph_reinstantiate.hip -- minimal HIP-only reproducer for the record gpurun_out/r2_qt1.log (DESIGN.md "r2_qt1"):
// capture N kernel launches into a graph, instantiate, launch, wait, DESTROY the exec, capture again (one kernel argument
// changed), instantiate, launch.  Runs clean on its own; the question is what it does under `rocprofv3 --kernel-trace`.
//   hipcc --offload-arch=gfx950 -O2 graph_reinstantiate.hip -o graph_reinstantiate
//   ./graph_reinstantiate [nodes=128] [keep_old_exec=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

struct Params { double* p; int n; int cap; double pad[24]; };     // ~220 bytes by value, like lpx::SelParams

__global__ void step(Params q)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < q.n && q.p[0] < q.cap) q.p[i] += 1.0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int build(hipStream_t s, const Params& q, int nodes, hipGraphExec_t* out)
{
    hipGraph_t g = nullptr;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) hipLaunchKernelGGL(step, dim3(64), dim3(256), 0, s, q);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    return 0;
}

int main(int argc, char** argv)
{
    const int nodes = argc > 1 ? std::atoi(argv[1]) : 128, keep = argc > 2 ? std::atoi(argv[2]) : 0;
    {   // executable mappings, so that a raw stack trace can be attributed to libraries
        std::ifstream m("/proc/self/maps"); std::string l;
        while (std::getline(m, l)) if (l.find("r-xp") != std::string::npos) std::cout << l << "\n";
    }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Params q{}; q.n = 64 * 256; q.cap = 2000;
    CK(hipMalloc((void**)&q.p, sizeof(double) * q.n));
    CK(hipMemset(q.p, 0, sizeof(double) * q.n));
    hipGraphExec_t a = nullptr, b = nullptr;
    if (build(s, q, nodes, &a)) return 1;
    for (int k = 0; k < 32; ++k) CK(hipGraphLaunch(a, s));
    CK(hipStreamSynchronize(s));
    std::printf("first graph: 32 launches done\n"); std::fflush(stdout);
    if (!keep) CK(hipGraphExecDestroy(a));
    q.cap = 10000;
    if (build(s, q, nodes, &b)) return 1;
    std::printf("second graph instantiated\n"); std::fflush(stdout);
    for (int k = 0; k < 32; ++k) CK(hipGraphLaunch(b, s));
    CK(hipStreamSynchronize(s));
    double v = 0; CK(hipMemcpy(&v, q.p + 1, sizeof(double), hipMemcpyDeviceToHost));
    std::printf("second graph: 32 launches done, p[1] = %.0f\n", v);
    return 0;
}
