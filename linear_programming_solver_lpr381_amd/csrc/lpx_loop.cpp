// lpx_loop.cpp -- error state, device binding and the generic device-resident loop driver.
#include "lpx_internal.h"

#include <chrono>
#include <cstring>
#include <cstdlib>
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
    std::call_once(g_init_once, [] { g_init_err = kernels_init(); if (g_init_err == hipSuccess) g_init_err = resident_init();
                                     if (g_init_err == hipSuccess) g_init_err = resident_group_init();
                                     if (g_init_err == hipSuccess) g_init_err = resident_regs_init();
                                     if (g_init_err == hipSuccess) g_init_err = resident_col_init(); });
    if (g_init_err != hipSuccess) {
        set_error(std::string("kernel attribute setup failed: ") + hipGetErrorString(g_init_err));
        return LPX_EDEVICE;
    }
    return 0;
}

hipStream_t borrow_stream()
{
    static constexpr int kStreams = 8;
    static hipStream_t pool[kStreams];
    static int made = 0, next = 0;
    if (made < kStreams) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return made ? pool[next++ % made] : nullptr;
        pool[made] = s;
        return pool[made++];
    }
    return pool[next++ % kStreams];
}

double now_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// Graph executables are not destroyed because a run came with another parameter set (another iteration cap, another
// forced-pivot list ...): the outgoing one is parked, keyed by its owner (the handle's gexec slot), batch and parameter bytes,
// and taken back when the same parameters return -- alternating parameter sets (bench legs, B&B depth classes, the revised
// path's drift-check segments) stop paying a capture + instantiate each time.  The park is bounded; the oldest entry is
// destroyed when it overflows, and an owner's entries go with the owner (graph_cache_drop_owner).  LPX_GRAPH_KEEP=0 restores
// destroy-at-once (diagnostic; it is NOT what the record r2_qt1 of DESIGN.md turned on: that abort shows with both settings).
namespace {
struct ParkedGraph { const void* owner; int batch; std::string key; hipGraphExec_t exec; };
std::mutex g_park_mu;
std::vector<ParkedGraph> g_park;                 // oldest first
constexpr size_t kParkMax = 96;
bool graph_keep()
{
    static const bool keep = [] { const char* e = std::getenv("LPX_GRAPH_KEEP"); return !(e && e[0] == '0'); }();
    return keep;
}
}  // namespace

void graph_cache_drop_owner(const void* owner)
{
    std::lock_guard<std::mutex> g(g_park_mu);
    for (size_t i = 0; i < g_park.size();) {
        if (g_park[i].owner == owner) { hipGraphExecDestroy(g_park[i].exec); g_park.erase(g_park.begin() + (long)i); }
        else ++i;
    }
}

static void retire_graph(LoopCtx& c)
{
    if (!*c.gexec) return;
    if (!graph_keep()) { hipGraphExecDestroy(*c.gexec); *c.gexec = nullptr; *c.g_batch = 0; return; }
    std::lock_guard<std::mutex> g(g_park_mu);
    if (g_park.size() >= kParkMax) { hipGraphExecDestroy(g_park.front().exec); g_park.erase(g_park.begin()); }
    g_park.push_back(ParkedGraph{(const void*)c.gexec, *c.g_batch, *c.g_key, *c.gexec});
    *c.gexec = nullptr; *c.g_batch = 0;
}

static bool take_parked_graph(LoopCtx& c, int batch)
{
    std::lock_guard<std::mutex> g(g_park_mu);
    for (size_t i = g_park.size(); i-- > 0;) {
        ParkedGraph& e = g_park[i];
        if (e.owner == (const void*)c.gexec && e.batch == batch && e.key == c.key) {
            *c.gexec = e.exec; *c.g_batch = batch; *c.g_key = c.key;
            g_park.erase(g_park.begin() + (long)i);
            return true;
        }
    }
    return false;
}

static int build_graph(LoopCtx& c, int batch)
{
    if (*c.gexec && *c.g_batch == batch && *c.g_key == c.key) return 0;
    retire_graph(c);
    if (take_parked_graph(c, batch)) return 0;
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

// ---------------------------------------------------------------------------------------------------
// LoopRun: iterations are enqueued batch-wise until the device state leaves LPX_RUNNING.  `budget`
// bounds the number of iterations ever enqueued (each one pivots, changes phase, or terminates).
// ---------------------------------------------------------------------------------------------------
int LoopRun::begin(const LoopCtx& c, const DevState& init, const lpx_run_opts* o, long long budget,
                   lpx_pivot_cb cb, void* user)
{
    c_ = c; o_ = *o; budget_ = budget; cb_ = cb; user_ = user;
    // pivots already made on this state: by another path (start_iter: resident hand-over) or by an earlier segment of the
    // same run (init.iter: revised path continuing after a refactorisation) -- their callbacks have been fired
    enq_ = 0; fired_ = init.iter > 0 ? init.iter : c.start_iter; status_ = LPX_RUNNING; init_phase_ = init.phase;
    std::memset(&local_, 0, sizeof(local_));
    batch_ = o_.batch > 0 ? o_.batch : 64;
    if (o_.profile && batch_ > 256) batch_ = 256;
    graph_ = o_.use_graph && !o_.profile;
    *c_.hst = init;                                 // pinned staging: safe for the async copy below
    if (c.start_iter > 0 && init.iter == 0) { c_.hst->iter = c.start_iter; c_.hst->primal_count = (init.phase == 2) ? c.start_iter : 0; }
    LPX_HIP_TRY(hipMemcpyAsync(c_.st, c_.hst, sizeof(DevState), hipMemcpyHostToDevice, c_.stream));
    LPX_HIP_TRY(hipStreamSynchronize(c_.stream));
    if (o_.profile) {
        size_t need = 2 * (size_t)batch_;
        while (c_.events->size() < need) {
            hipEvent_t e;
            LPX_HIP_TRY(hipEventCreate(&e));
            c_.events->push_back(e);
        }
    }
    if (graph_) { int rc = build_graph(c_, batch_); if (rc) return rc; }
    if (!stage_) LPX_HIP_TRY(hipHostMalloc((void**)&stage_, 2 * sizeof(DevState)));
    for (int k = 0; k < 2; ++k) if (!ev_[k]) LPX_HIP_TRY(hipEventCreateWithFlags(&ev_[k], hipEventDisableTiming));
    head_ = tail_ = 0;
    t0_ = now_ms();
    if (c_.prologue) { int rc = c_.prologue(c_.stream); if (rc) return rc; local_.launches += 1; }
    return 0;
}

int LoopRun::submit()
{
    iter_before_ = fired_;
    if (graph_) {
        LPX_HIP_TRY(hipGraphLaunch(*c_.gexec, c_.stream));
    } else {
        for (int i = 0; i < batch_; ++i) {
            int rc = o_.profile ? c_.enqueue_iter(c_.stream, (*c_.events)[2 * i], (*c_.events)[2 * i + 1])
                                : c_.enqueue_iter(c_.stream, nullptr, nullptr);
            if (rc) return rc;
        }
    }
    enq_ += batch_;
    local_.launches += (long long)c_.launches_per_iter * batch_;
    if (head_ - tail_ >= 2) { set_error("LoopRun: more than two batches in flight"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipMemcpyAsync(&stage_[head_ & 1], c_.st, sizeof(DevState), hipMemcpyDeviceToHost, c_.stream));
    LPX_HIP_TRY(hipEventRecord(ev_[head_ & 1], c_.stream));
    ++head_;
    return 0;
}

LoopRun::~LoopRun()
{
    if (head_ > tail_ && c_.stream) hipStreamSynchronize(c_.stream);     // a copy into stage_ may still be in flight
    if (stage_) hipHostFree(stage_);
    for (int k = 0; k < 2; ++k) if (ev_[k]) hipEventDestroy(ev_[k]);
}

int LoopRun::complete()
{
    if (head_ == tail_) { set_error("LoopRun::complete without a batch in flight"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipEventSynchronize(ev_[tail_ & 1]));
    *c_.hst = stage_[tail_ & 1];
    ++tail_;
    status_ = c_.hst->status;
    const int done = c_.hst->iter;
    if (o_.profile && c_.profile_maps) {
        // the first (done - iter_before) iterations of this batch each ran one full update
        const int full = done - iter_before_;
        for (int i = 0; i < full && i < batch_; ++i) {
            float ms = 0.f;
            LPX_HIP_TRY(hipEventElapsedTime(&ms, (*c_.events)[2 * i], (*c_.events)[2 * i + 1]));
            local_.update_ms_sum += ms;
            local_.update_launches++;
        }
    }
    if (cb_ && done > fired_) {
        int lo = fired_, hi = done < c_.trace_cap ? done : c_.trace_cap;
        if (hi > lo) {
            std::vector<int32_t> tr(2 * (size_t)(hi - lo));
            LPX_HIP_TRY(hipMemcpy(tr.data(), c_.trace + 2 * lo, sizeof(int32_t) * 2 * (hi - lo), hipMemcpyDeviceToHost));
            for (int k = lo; k < hi; ++k) cb_(user_, k + 1, tr[2 * (k - lo)], tr[2 * (k - lo) + 1]);
        }
    }
    fired_ = done;
    return 0;
}

int LoopRun::finish(lpx_stats* stats)
{
    local_.loop_ms = now_ms() - t0_;
    local_.pivots = c_.hst->iter;
    local_.fdf_pivots = c_.hst->fdf_count;
    local_.cleanup_pivots = (init_phase_ == 0) ? c_.hst->primal_count : 0;
    if (stats) {
        double h2d = stats->h2d_ms, d2h = stats->d2h_ms;
        *stats = local_; stats->h2d_ms = h2d; stats->d2h_ms = d2h;
    }
    if (status_ == LPX_RUNNING) { set_error("loop budget exhausted while still running"); return LPX_ITER_LIMIT; }
    return status_;
}

int run_device_loop(LoopCtx& c, const DevState& init, const lpx_run_opts* o, long long budget,
                    lpx_pivot_cb cb, void* user, lpx_stats* stats)
{
    LoopRun run;
    int rc = run.begin(c, init, o, budget, cb, user);
    if (rc) return rc;
    if (o->profile || cb) {                 // event-bracketed launches, or a callback that may look at the tableau: one batch at a time
        while (!run.done()) {
            rc = run.submit(); if (rc) return rc;
            rc = run.complete(); if (rc) return rc;
        }
        return run.finish(stats);
    }
    // One batch AHEAD: batch k + 1 is queued before the host waits for batch k, so the device never idles through the
    // host's poll (state copy, wake-up, graph launch: 40-60 us per batch on a quiet host, ten times that on a busy one --
    // a bench run lost 7 % to it).  When the loop ends inside batch k the batch ahead finds the state record finished and
    // every kernel of it leaves at once.
    if (!run.done()) { rc = run.submit(); if (rc) return rc; }
    while (run.in_flight() > 0) {
        if (run.may_submit() && run.in_flight() < 2 && !run.done()) { rc = run.submit(); if (rc) return rc; }
        rc = run.complete(); if (rc) return rc;
        if (run.done()) {                   // finished (or budget spent): drain what is still queued, enqueue nothing new
            while (run.in_flight() > 0) { rc = run.complete(); if (rc) return rc; }
            break;
        }
    }
    return run.finish(stats);
}

}  // namespace lpx
