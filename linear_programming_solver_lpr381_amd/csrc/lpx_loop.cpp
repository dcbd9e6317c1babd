// lpx_loop.cpp -- error state, device binding and the generic device-resident loop driver.
#include "lpx_internal.h"

#include <chrono>
#include <cstring>
#include <mutex>

namespace lpx {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const std::string& get_error() { return g_err; }

static std::once_flag g_init_once;
static hipError_t g_init_err = hipSuccess;
int g_device = -1;

int ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device visible (liblpx has no CPU fallback)");
        return LPX_EDEVICE;
    }
    if (g_device < 0) {
        g_device = 0;
        LPX_HIP_TRY(hipSetDevice(0));
    }
    std::call_once(g_init_once, [] { g_init_err = kernels_init(); });
    if (g_init_err != hipSuccess) {
        set_error(std::string("kernel attribute setup failed: ") + hipGetErrorString(g_init_err));
        return LPX_EDEVICE;
    }
    return 0;
}

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static void drop_graph(LoopCtx& c)
{
    if (*c.gexec) { hipGraphExecDestroy(*c.gexec); *c.gexec = nullptr; *c.g_batch = 0; }
}

static int build_graph(LoopCtx& c, int batch)
{
    if (*c.gexec && *c.g_batch == batch && *c.g_key == c.key) return 0;
    drop_graph(c);
    hipGraph_t graph = nullptr;
    LPX_HIP_TRY(hipStreamBeginCapture(c.stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < batch; ++i) {
        int rc = c.enqueue_iter(c.stream, nullptr, nullptr);
        if (rc) { hipStreamEndCapture(c.stream, &graph); if (graph) hipGraphDestroy(graph); return rc; }
    }
    LPX_HIP_TRY(hipStreamEndCapture(c.stream, &graph));
    hipError_t e = hipGraphInstantiate(c.gexec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) { *c.gexec = nullptr; set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); return LPX_EDEVICE; }
    *c.g_batch = batch;
    *c.g_key = c.key;
    return 0;
}

// Runs iterations until the device state leaves LPX_RUNNING.  `budget` bounds the number of
// iterations ever enqueued (each one either pivots, changes phase, or terminates).
int run_device_loop(LoopCtx& c, const DevState& init, const lpx_run_opts* o, long long budget,
                    lpx_pivot_cb cb, void* user, lpx_stats* stats)
{
    int batch = o->batch > 0 ? o->batch : 64;
    if (o->profile && batch > 256) batch = 256;
    const bool graph = o->use_graph && !o->profile;
    lpx_stats local; std::memset(&local, 0, sizeof(local));

    LPX_HIP_TRY(hipMemcpyAsync(c.st, &init, sizeof(init), hipMemcpyHostToDevice, c.stream));
    LPX_HIP_TRY(hipStreamSynchronize(c.stream));

    if (o->profile) {
        size_t need = 2 * (size_t)batch;
        while (c.events->size() < need) {
            hipEvent_t e;
            LPX_HIP_TRY(hipEventCreate(&e));
            c.events->push_back(e);
        }
    }
    if (graph) { int rc = build_graph(c, batch); if (rc) return rc; }

    const double t0 = now_ms();
    if (c.prologue) { int rc = c.prologue(c.stream); if (rc) return rc; local.launches += 1; }
    int fired = 0;
    long long enq = 0;
    int status = LPX_RUNNING;
    while (status == LPX_RUNNING && enq < budget) {
        const int iter_before = fired;
        if (graph) {
            LPX_HIP_TRY(hipGraphLaunch(*c.gexec, c.stream));
        } else {
            for (int i = 0; i < batch; ++i) {
                int rc = o->profile ? c.enqueue_iter(c.stream, (*c.events)[2 * i], (*c.events)[2 * i + 1])
                                    : c.enqueue_iter(c.stream, nullptr, nullptr);
                if (rc) return rc;
            }
        }
        enq += batch;
        local.launches += (long long)c.launches_per_iter * batch;
        LPX_HIP_TRY(hipMemcpyAsync(c.hst, c.st, sizeof(DevState), hipMemcpyDeviceToHost, c.stream));
        LPX_HIP_TRY(hipStreamSynchronize(c.stream));
        status = c.hst->status;
        const int done = c.hst->iter;
        if (o->profile && c.profile_maps) {
            // the first (done - iter_before) iterations of this batch each ran one full update
            const int full = done - iter_before;
            for (int i = 0; i < full && i < batch; ++i) {
                float ms = 0.f;
                LPX_HIP_TRY(hipEventElapsedTime(&ms, (*c.events)[2 * i], (*c.events)[2 * i + 1]));
                local.update_ms_sum += ms;
                local.update_launches++;
            }
        }
        if (cb && done > fired) {
            int lo = fired, hi = done < c.trace_cap ? done : c.trace_cap;
            if (hi > lo) {
                std::vector<int32_t> tr(2 * (size_t)(hi - lo));
                LPX_HIP_TRY(hipMemcpy(tr.data(), c.trace + 2 * lo, sizeof(int32_t) * 2 * (hi - lo), hipMemcpyDeviceToHost));
                for (int k = lo; k < hi; ++k) cb(user, k + 1, tr[2 * (k - lo)], tr[2 * (k - lo) + 1]);
            }
        }
        fired = done;
    }
    local.loop_ms = now_ms() - t0;
    local.pivots = c.hst->iter;
    local.fdf_pivots = c.hst->fdf_count;
    local.cleanup_pivots = (init.phase == 0) ? c.hst->primal_count : 0;
    if (stats) {
        double h2d = stats->h2d_ms, d2h = stats->d2h_ms;
        *stats = local; stats->h2d_ms = h2d; stats->d2h_ms = d2h;
    }
    if (status == LPX_RUNNING) { set_error("loop budget exhausted while still running"); return LPX_ITER_LIMIT; }
    return status;
}

}  // namespace lpx
