// lpx_tableau.cpp -- device-resident tableau handle and the host side of the simplex loops
// (C ABI of include/lpx.h).  Host code only: kernels live in lpx_kernels.hip.
#include "lpx_internal.h"

#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace lpx { const std::string& get_error(); extern int g_device; }

using namespace lpx;

struct lpx_tableau {
    int R = 0, C = 0, ld = 0;   // live shape (<= capacity) and leading dimension (from the capacity)
    int Rcap = 0, Ccap = 0;
    char* slab = nullptr;       // device slab holding every small buffer below (all but T)
    char* hslab = nullptr;      // pinned slab holding hst and shape_h
    int32_t* shape = nullptr;   // device record {R, C} read by the kernels
    int32_t* shape_h = nullptr; // pinned staging
    double* T = nullptr;        // [R*ld]
    double* snapT = nullptr;    // snapshot
    double* prow = nullptr;     // [ld]
    double* pcol = nullptr;     // [R]
    double* col0 = nullptr;     // [R] lookahead column buffers (ping-pong)
    double* col1 = nullptr;
    double* rhsbuf = nullptr;   // [R]
    double* ws = nullptr;       // [MB_MAXB * max(R,C)]
    double* part_v = nullptr; int32_t* part_i = nullptr;   // [64] partial argmins of the multi-workgroup select
    DevState* us = nullptr;     // state record written by the update kernel (multi-workgroup protocol)
    int use_mb = 1;
    int32_t* basis = nullptr;   // [R-1]
    int32_t* snapBasis = nullptr;
    int32_t* trace = nullptr;   // [2*trace_cap]
    int trace_cap = 0;
    DevState* st = nullptr;     // device
    DevState* hst = nullptr;    // pinned host mirror
    bool suspended = false;     // lpx_multi_run_some left this run unfinished: *hst is where it continues
    int32_t* frows = nullptr; int32_t* fcols = nullptr; int32_t* fchosen = nullptr; int fcap = 0;
    char* cutbuf = nullptr; char* cutbuf_h = nullptr; int cutcap = 0;   // staging of branching-row descriptors
    hipStream_t stream = nullptr;
    // cached graph of `g_batch` (select, update) pairs
    hipGraphExec_t gexec = nullptr;
    int g_batch = 0;
    std::string g_key;
    std::vector<hipEvent_t> events;
    // resident primal loop: exchange buffers (tagged granules) and the generation counter
    unsigned long long* xr = nullptr; unsigned long long* xp = nullptr; unsigned* xgen = nullptr;
    int32_t* xbasis = nullptr;      // basis as it was when the current resident launch started
    double* xT = nullptr;           // tableau as it was when the current resident launch started (put back if the launch aborts)
    unsigned long long* xc = nullptr; unsigned long long* xq = nullptr;   // column-owning resident kernel: candidates / candidate columns
    size_t xc_bytes = 0, xq_bytes = 0;
    bool resident_off = false;      // a resident launch could not get its workgroups co-resident: stay on the streaming path
    // fused pivot (lpx_pivot_fused): second tableau buffer and the index-1 copies of the small per-pivot vectors, on first use
    double* fT = nullptr; char* fslab = nullptr;
    double* fprow = nullptr; double* frhs = nullptr; DevState* frec = nullptr;
    bool fused_off = false;         // the second buffer did not fit: stay on the two-launch path
    bool suspended2 = false;        // ... by the two-launch group kernels (it must continue there: no pending pivot, state in *hst)
    bool fsuspended = false; int frec_cur = 0;   // fused group run left unfinished: its records (latest: index frec_cur) are in place
};

static constexpr int LPX_RESIDENT_RETRY = -1000;     // internal: first resident launch timed out, state untouched

static void drop_graph(lpx_tableau* t)
{
    if (t->gexec) { hipGraphExecDestroy(t->gexec); t->gexec = nullptr; t->g_batch = 0; }
    graph_cache_drop_owner(&t->gexec);          // the parked ones captured the same buffers
}

extern "C" {

int lpx_abi_version(void) { return LPX_ABI_VERSION; }

int lpx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lpx_init(int device)
{
    int n = lpx_device_count();
    if (device < 0 || device >= n) { set_error("lpx_init: device index out of range"); return LPX_EDEVICE; }
    LPX_HIP_TRY(hipSetDevice(device));
    g_device = device;
    return ensure_device();
}

int lpx_last_error(char* buf, int len)
{
    const std::string& e = get_error();
    if (!buf || len <= 0) return (int)e.size();
    std::strncpy(buf, e.c_str(), len - 1);
    buf[len - 1] = 0;
    return (int)e.size();
}

int lpx_device_name(char* buf, int len)
{
    int rc = ensure_device();
    if (rc) return rc;
    hipDeviceProp_t prop;
    LPX_HIP_TRY(hipGetDeviceProperties(&prop, g_device));
    std::snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

void lpx_default_opts(lpx_run_opts* o, int dual)
{
    std::memset(o, 0, sizeof(*o));
    o->eps = 1e-9;
    o->ratio_tol = dual ? 1e-12 : 1e-9;
    o->max_iter = 10000;
    o->fdf_guard = 100;
    o->cleanup = 0;
    o->batch = 0;
    static const bool no_graph = [] { const char* e = std::getenv("LPX_GRAPH"); return e && e[0] == '0'; }();
    o->use_graph = no_graph ? 0 : 1;     // LPX_GRAPH=0: diagnostic switch to eager launches
    o->profile = 0;
    o->resident = 0;
}

int lpx_tableau_create(int R, int C, lpx_tableau** out)
{
    if (!out || R < 1 || C < 2) { set_error("lpx_tableau_create: bad shape"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    lpx_tableau* t = new lpx_tableau();
    t->R = R; t->C = C; t->Rcap = R; t->Ccap = C; t->ld = (C + 15) / 16 * 16;
    const size_t tb = sizeof(double) * (size_t)R * t->ld;
    const int wsn = R > C ? R : C;
    t->trace_cap = 1 << 16;
    // One device slab for the tableau and one for everything small around it, one pinned slab for the host mirrors:
    // a B&B pool creates dozens of handles per solve, and 14 hipMallocs + 2 hipHostMallocs each cost 4 ms per handle.
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t nb = R > 1 ? (size_t)R - 1 : 1;
    const size_t sz[] = { up(sizeof(double) * t->ld), up(sizeof(double) * R), up(sizeof(double) * R), up(sizeof(double) * R),
                          up(sizeof(double) * R), up(sizeof(double) * (size_t)wsn * 32), up(sizeof(double) * 192),
                          up(sizeof(int32_t) * 256), up(sizeof(DevState)), up(sizeof(int32_t) * nb),
                          up(sizeof(int32_t) * 2 * (size_t)t->trace_cap), up(sizeof(DevState)), up(sizeof(int32_t) * 2) };
    size_t total = 0;
    for (size_t b : sz) total += b;
    hipError_t e_ = malloc_retry((void**)&t->T, tb);
    if (e_ == hipSuccess) e_ = malloc_retry((void**)&t->slab, total);
    if (e_ != hipSuccess) {
        set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e_));
        lpx_tableau_destroy(t);
        return e_ == hipErrorOutOfMemory ? LPX_ENOMEM : LPX_EDEVICE;
    }
    {
        char* p = t->slab; int k = 0;
        t->prow = (double*)p; p += sz[k++];
        t->pcol = (double*)p; p += sz[k++];
        t->col0 = (double*)p; p += sz[k++];
        t->col1 = (double*)p; p += sz[k++];
        t->rhsbuf = (double*)p; p += sz[k++];
        t->ws = (double*)p; p += sz[k++];
        t->part_v = (double*)p; p += sz[k++];      // [128..191]: diagnostic stamps (LPX_STAMPS builds only)
        t->part_i = (int32_t*)p; p += sz[k++];
        t->us = (DevState*)p; p += sz[k++];
        t->basis = (int32_t*)p; p += sz[k++];
        t->trace = (int32_t*)p; p += sz[k++];
        t->st = (DevState*)p; p += sz[k++];
        t->shape = (int32_t*)p; p += sz[k++];
    }
    if (hipHostMalloc((void**)&t->hslab, 256 + sizeof(int32_t) * 2) != hipSuccess ||
        (t->stream = borrow_stream()) == nullptr) {
        set_error("host-pinned state / stream creation failed");
        lpx_tableau_destroy(t);
        return LPX_EDEVICE;
    }
    t->hst = (DevState*)t->hslab; t->shape_h = (int32_t*)(t->hslab + 256);
    static_assert(sizeof(DevState) <= 256, "pinned slab layout");
    hipMemsetAsync(t->T, 0, tb, t->stream);
    hipMemsetAsync(t->slab, 0, total - sz[10] - sz[11] - sz[12], t->stream);      // everything in front of the trace
    hipMemsetAsync(t->st, 0, sizeof(DevState), t->stream);
    t->shape_h[0] = R; t->shape_h[1] = C;
    hipMemcpyAsync(t->shape, t->shape_h, sizeof(int32_t) * 2, hipMemcpyHostToDevice, t->stream);
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    *out = t;
    return 0;
}

void lpx_tableau_destroy(lpx_tableau* t)
{
    if (!t) return;
    if (t->stream) hipStreamSynchronize(t->stream);
    drop_graph(t);
    for (hipEvent_t e : t->events) hipEventDestroy(e);
    hipFree(t->T); hipFree(t->slab); hipFree(t->snapT); hipFree(t->snapBasis);
    hipFree(t->frows); hipFree(t->fcols); hipFree(t->fchosen); hipFree(t->cutbuf);
    hipFree(t->fT); hipFree(t->fslab);
    hipFree(t->xr); hipFree(t->xp); hipFree(t->xgen); hipFree(t->xbasis); hipFree(t->xT); hipFree(t->xc); hipFree(t->xq);
    if (t->hslab) hipHostFree(t->hslab);
    if (t->cutbuf_h) hipHostFree(t->cutbuf_h);
    delete t;                       // the stream is borrowed (borrow_stream), not owned
}

int lpx_tableau_shape(const lpx_tableau* t, int* R, int* C, int* ld)
{
    if (!t) return LPX_EINVAL;
    if (R) *R = t->R;
    if (C) *C = t->C;
    if (ld) *ld = t->ld;
    return 0;
}

int lpx_tableau_upload(lpx_tableau* t, const double* T, const int32_t* basis)
{
    if (!t || !T) { set_error("lpx_tableau_upload: null argument"); return LPX_EINVAL; }
    t->suspended = t->suspended2 = t->fsuspended = false;
    LPX_HIP_TRY(hipMemcpy2DAsync(t->T, sizeof(double) * t->ld, T, sizeof(double) * t->C,
                                 sizeof(double) * t->C, t->R, hipMemcpyHostToDevice, t->stream));
    if (basis && t->R > 1)
        LPX_HIP_TRY(hipMemcpyAsync(t->basis, basis, sizeof(int32_t) * (t->R - 1), hipMemcpyHostToDevice, t->stream));
    LPX_HIP_TRY(hipMemsetAsync(t->st, 0, sizeof(DevState), t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int lpx_tableau_download(lpx_tableau* t, double* T, int32_t* basis)
{
    if (!t) return LPX_EINVAL;
    if (T)
        LPX_HIP_TRY(hipMemcpy2DAsync(T, sizeof(double) * t->C, t->T, sizeof(double) * t->ld,
                                     sizeof(double) * t->C, t->R, hipMemcpyDeviceToHost, t->stream));
    if (basis && t->R > 1)
        LPX_HIP_TRY(hipMemcpyAsync(basis, t->basis, sizeof(int32_t) * (t->R - 1), hipMemcpyDeviceToHost, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int lpx_tableau_snapshot(lpx_tableau* t)
{
    if (!t) return LPX_EINVAL;
    const size_t tb = sizeof(double) * (size_t)t->R * t->ld;
    if (!t->snapT) {
        LPX_HIP_TRY(hipMalloc((void**)&t->snapT, tb));
        LPX_HIP_TRY(hipMalloc((void**)&t->snapBasis, sizeof(int32_t) * (t->R > 1 ? t->R - 1 : 1)));
    }
    LPX_HIP_TRY(hipMemcpyAsync(t->snapT, t->T, tb, hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipMemcpyAsync(t->snapBasis, t->basis, sizeof(int32_t) * (t->R > 1 ? t->R - 1 : 1),
                               hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int lpx_tableau_restore(lpx_tableau* t)
{
    if (!t || !t->snapT) { set_error("lpx_tableau_restore: no snapshot"); return LPX_EINVAL; }
    t->suspended = t->suspended2 = t->fsuspended = false;
    const size_t tb = sizeof(double) * (size_t)t->R * t->ld;
    LPX_HIP_TRY(hipMemcpyAsync(t->T, t->snapT, tb, hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipMemcpyAsync(t->basis, t->snapBasis, sizeof(int32_t) * (t->R > 1 ? t->R - 1 : 1),
                               hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipMemsetAsync(t->st, 0, sizeof(DevState), t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int lpx_tableau_device_ptr(lpx_tableau* t, void** dptr, int* ld)
{
    if (!t) return LPX_EINVAL;
    if (dptr) *dptr = t->T;
    if (ld) *ld = t->ld;
    return 0;
}

#ifdef LPX_STAMPS
int lpx_debug_resident_col(lpx_tableau* t, unsigned long long* out, int n, int clear)
{
    if (!t->xc) return LPX_EINVAL;
    LPX_HIP_TRY(hipMemcpy(out, t->xc + 2 * 256 * 2 * 2, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    if (clear) LPX_HIP_TRY(hipMemset(t->xc + 2 * 256 * 2 * 2, 0, sizeof(unsigned long long) * n));
    return 0;
}
int lpx_debug_resident_group(lpx_tableau* t, unsigned long long* out, int n, int clear)
{
    if (!t->xp) return LPX_EINVAL;
    LPX_HIP_TRY(hipMemcpy(out, t->xp + 4 * ((size_t)t->ld + 8), sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    if (clear) LPX_HIP_TRY(hipMemset(t->xp + 4 * ((size_t)t->ld + 8), 0, sizeof(unsigned long long) * n));
    return 0;
}
int lpx_debug_resident(lpx_tableau* t, unsigned long long* out, int n, int clear)
{
    if (!t->xp) return LPX_EINVAL;
    LPX_HIP_TRY(hipMemcpy(out, t->xp + 4 * (size_t)t->ld, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    if (clear) LPX_HIP_TRY(hipMemset(t->xp + 4 * (size_t)t->ld, 0, sizeof(unsigned long long) * n));
    return 0;
}
#endif
#ifdef LPX_STAMPS
extern "C++" { namespace lpx { hipError_t debug_copy_stamps(unsigned long long* out, int clear); } }
int lpx_debug_hs(unsigned long long* out, int clear) { LPX_HIP_TRY(lpx::debug_copy_stamps(out, clear)); return 0; }
int lpx_debug_ws(lpx_tableau* t, unsigned long long* out, int n, int clear)
{
    double* src = t->us ? t->part_v + 128 : t->ws;
    LPX_HIP_TRY(hipMemcpy(out, src, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
    if (clear) LPX_HIP_TRY(hipMemset(src, 0, sizeof(unsigned long long) * n));
    return 0;
}
#endif

int lpx_tableau_trace(lpx_tableau* t, int32_t* trace, int cap, int* n)
{
    if (!t) return LPX_EINVAL;
    LPX_HIP_TRY(hipMemcpyAsync(t->hst, t->st, sizeof(DevState), hipMemcpyDeviceToHost, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    int k = t->hst->iter;
    if (k > t->trace_cap) k = t->trace_cap;
    if (n) *n = k;
    if (trace && cap > 0) {
        int c = k < cap ? k : cap;
        if (c > 0) LPX_HIP_TRY(hipMemcpy(trace, t->trace, sizeof(int32_t) * 2 * c, hipMemcpyDeviceToHost));
    }
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// tableau loops on top of the generic driver (lpx_loop.cpp)
// ---------------------------------------------------------------------------------------------------
namespace {

int enqueue_pair(lpx_tableau* t, const SelParams& p, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr)
{
    if (p.mode == MODE_DUAL) {
        LPX_HIP_TRY(launch_select(p, s));
        LPX_HIP_TRY(launch_update(t->T, t->ld, t->Rcap, t->Ccap, t->shape, t->prow, t->pcol, t->pcol, t->rhsbuf, t->st, s, e0, e1));
    } else if (p.us) {
        LPX_HIP_TRY(launch_select_mb(p, s));
        LPX_HIP_TRY(launch_update_mb(p, s, e0, e1));
    } else {
        LPX_HIP_TRY(launch_select_la(p, s));
        LPX_HIP_TRY(launch_update(t->T, t->ld, t->Rcap, t->Ccap, t->shape, t->prow, t->col0, t->col1, t->rhsbuf, t->st, s, e0, e1));
    }
    return 0;
}

void make_ctx(lpx_tableau* t, const SelParams& p, LoopCtx& c, DevState& init)
{
    c.stream = t->stream; c.st = t->st; c.hst = t->hst; c.trace = t->trace; c.trace_cap = t->trace_cap;
    c.events = &t->events; c.gexec = &t->gexec; c.g_batch = &t->g_batch; c.g_key = &t->g_key;
    c.key.assign(reinterpret_cast<const char*>(&p), sizeof(p));
    c.enqueue_iter = [t, p](hipStream_t s, hipEvent_t e0, hipEvent_t e1) -> int { return enqueue_pair(t, p, s, e0, e1); };
    if (p.mode != MODE_DUAL)           // lookahead path: first entering column + its gather, once
        c.prologue = [p](hipStream_t s) -> int { LPX_HIP_TRY(launch_la_init(p, s)); return 0; };
    else                               // dual path: contiguous copy of the RHS column, once
        c.prologue = [p](hipStream_t s) -> int { LPX_HIP_TRY(launch_rhs_init(p, s)); return 0; };
    c.launches_per_iter = 2;
    c.profile_maps = (p.mode != MODE_DUAL);     // phase hops make the mapping ambiguous in dual mode
    std::memset(&init, 0, sizeof(init));
    init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1;
    init.phase = (p.mode == MODE_DUAL) ? 0 : 2;
}

int run_loop(lpx_tableau* t, SelParams p, const lpx_run_opts* o, long long budget,
             lpx_pivot_cb cb, void* user, lpx_stats* stats, int start_iter = 0)
{
    LoopCtx c; DevState init;
    make_ctx(t, p, c, init);
    c.start_iter = start_iter;
    return run_device_loop(c, init, o, budget, cb, user, stats);
}

SelParams base_params(lpx_tableau* t, const lpx_run_opts* o, int mode)
{
    SelParams p; std::memset(&p, 0, sizeof(p));
    p.T = t->T; p.ld = t->ld; p.R = t->Rcap; p.C = t->Ccap; p.shape = t->shape;
    p.prow = t->prow; p.pcol = t->pcol; p.col0 = t->col0; p.col1 = t->col1; p.rhsbuf = t->rhsbuf; p.basis = t->basis; p.trace = t->trace; p.trace_cap = t->trace_cap;
    p.st = t->st;
    p.eps = o->eps;
    p.tol_fdf = o->ratio_tol; p.tol_dual = o->ratio_tol;
    p.tol_primal = (mode == MODE_DUAL) ? o->eps : o->ratio_tol;
    p.max_iter = o->max_iter; p.fdf_guard = o->fdf_guard; p.cleanup = o->cleanup; p.mode = mode;
    p.ws = t->ws;
    p.rcap = 0;
    static const bool mb_env = [] { const char* e = std::getenv("LPX_SELECT_MB"); return !(e && e[0] == '0'); }();
    if (mode != MODE_DUAL && mb_env && t->use_mb) {
        p.us = t->us; p.part_v = t->part_v; p.part_i = t->part_i; p.nblk = select_mb_blocks(t->Ccap); p.qsel = update_policy(t->ld, t->Rcap) != 0 ? 1 : 0;
    }
    return p;
}

// Fused pivot: buffers on first use.  Returns false (and remembers it) when the second tableau does not fit the device.
static bool fused_buffers(lpx_tableau* t)
{
    if (t->fT) return true;
    if (t->fused_off) return false;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t tb = sizeof(double) * (size_t)t->Rcap * t->ld;
    const size_t sz[] = { up(sizeof(double) * t->ld), up(sizeof(double) * t->Rcap), up(2 * sizeof(DevState)) };
    if (malloc_retry((void**)&t->fT, tb) != hipSuccess || malloc_retry((void**)&t->fslab, sz[0] + sz[1] + sz[2]) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(t->fT); hipFree(t->fslab);
        t->fT = nullptr; t->fslab = nullptr; t->fused_off = true;
        return false;
    }
    t->fprow = (double*)t->fslab;
    t->frhs = (double*)(t->fslab + sz[0]);
    t->frec = (DevState*)(t->fslab + sz[0] + sz[1]);
    hipMemsetAsync(t->fT, 0, tb, t->stream);
    hipMemsetAsync(t->fslab, 0, sz[0] + sz[1] + sz[2], t->stream);
    return true;
}

// Primal loop with ONE launch per pivot (lpx_pivot_fused): update(k) out of place beside select(k+1).  The state record the
// host polls is one launch behind the device's, so the loop gets a few iterations of slack; when it ends the tableau may sit
// in the second buffer and is brought home (a device-to-device copy of the live rows, ~0.1 ms per 400 MB, once per solve).
static int run_fused(lpx_tableau* t, const SelParams& p, const lpx_run_opts* o, lpx_stats* stats, int start_iter)
{
    FusedParams f; std::memset(&f, 0, sizeof(f));
    f.P = p; f.T1 = t->fT; f.prow1 = t->fprow; f.rhs1 = t->frhs; f.rec = t->frec;
    LoopCtx c; DevState init;
    make_ctx(t, p, c, init);
    c.key.assign(reinterpret_cast<const char*>(&f), sizeof(f));
    // A launch reads state record `par` and writes the other one; par alternates from launch to launch.  The prologue's launch is
    // number 0, so the loop proper starts at 1 -- in the captured graph too (it is captured before the prologue runs, hence the
    // counter is preset), and a graph batch is made even so that every replay starts at the parity the capture started at.
    auto count = std::make_shared<long long>(1);
    c.enqueue_iter = [f, count](hipStream_t s, hipEvent_t e0, hipEvent_t e1) -> int {
        LPX_HIP_TRY(launch_pivot_fused(f, (int)(*count & 1), s, e0, e1)); ++*count; return 0; };
    // the prologue also selects the first pivot, so that every launch of the loop proper has a pivot to apply
    c.prologue = [f, count](hipStream_t s) -> int {
        LPX_HIP_TRY(launch_fused_init(f, s)); LPX_HIP_TRY(launch_pivot_fused(f, 0, s)); *count = 1; return 0; };
    c.launches_per_iter = 1;
    c.start_iter = start_iter;
    lpx_run_opts oe = *o;
    { const int b = oe.batch > 0 ? oe.batch : 64; oe.batch = (b + 1) & ~1; }
    o = &oe;
    const int rc = run_device_loop(c, init, o, (long long)o->max_iter + 4, nullptr, nullptr, stats);
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    DevState recs[2];
    LPX_HIP_TRY(hipMemcpy(recs, t->frec, sizeof(recs), hipMemcpyDeviceToHost));
    const DevState& last = recs[1].pad[2] > recs[0].pad[2] ? recs[1] : recs[0];
    if (last.pad[3] == 1) {
        LPX_HIP_TRY(hipMemcpyAsync(t->T, t->fT, sizeof(double) * (size_t)t->R * t->ld, hipMemcpyDeviceToDevice, t->stream));
        LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    }
    return rc;
}

// Exchange buffers of the resident kernels (sized for both of them) and the basis snapshot, allocated on first use.
static size_t xr_bytes(const lpx_tableau* t) { return sizeof(unsigned long long) * 8 * (size_t)t->Rcap; }
static size_t xp_bytes(const lpx_tableau* t) { return sizeof(unsigned long long) * (4 * ((size_t)t->ld + 8) + 64); }   // + diagnostic stamps
static int resident_buffers(lpx_tableau* t)
{
    if (t->xr) return 0;
    LPX_HIP_TRY(hipMalloc((void**)&t->xr, xr_bytes(t)));
    LPX_HIP_TRY(hipMalloc((void**)&t->xp, xp_bytes(t)));
    LPX_HIP_TRY(hipMalloc((void**)&t->xgen, sizeof(unsigned)));
    LPX_HIP_TRY(hipMalloc((void**)&t->xbasis, sizeof(int32_t) * (size_t)t->Rcap));
    LPX_HIP_TRY(malloc_retry((void**)&t->xT, sizeof(double) * (size_t)t->Rcap * t->ld));
    LPX_HIP_TRY(hipMemsetAsync(t->xr, 0, xr_bytes(t), t->stream));
    LPX_HIP_TRY(hipMemsetAsync(t->xp, 0, xp_bytes(t), t->stream));
    LPX_HIP_TRY(hipMemsetAsync(t->xgen, 0, sizeof(unsigned), t->stream));
    return 0;
}
// exchange buffers of the column-owning kernel, sized for the handle's capacity and every grid up to the CU count
static int resident_col_buffers(lpx_tableau* t, int grid)
{
    const size_t xcb = resident_col_xc_bytes(grid > 256 ? grid : 256), xqb = resident_col_xq_bytes(grid > 256 ? grid : 256, t->Rcap);
    if (t->xc && t->xc_bytes >= xcb && t->xq_bytes >= xqb) return 0;
    hipFree(t->xc); hipFree(t->xq); t->xc = nullptr; t->xq = nullptr;
    LPX_HIP_TRY(hipMalloc((void**)&t->xc, xcb));
    LPX_HIP_TRY(hipMalloc((void**)&t->xq, xqb));
    t->xc_bytes = xcb; t->xq_bytes = xqb;
    LPX_HIP_TRY(hipMemsetAsync(t->xc, 0, xcb, t->stream));
    LPX_HIP_TRY(hipMemsetAsync(t->xq, 0, xqb, t->stream));
    return 0;
}
static void resident_buffers_clear(lpx_tableau* t)
{
    if (t->xc) { hipMemsetAsync(t->xc, 0, t->xc_bytes, t->stream); hipMemsetAsync(t->xq, 0, t->xq_bytes, t->stream); }
    hipMemsetAsync(t->xr, 0, xr_bytes(t), t->stream);
    hipMemsetAsync(t->xp, 0, xp_bytes(t), t->stream);
    hipStreamSynchronize(t->stream);
}

// Resident primal loop: one launch runs up to `chunk` pivots with the tableau in LDS; the host only polls the
// 64-byte state record between launches (and fires the pivot callbacks from the trace).
// col: the column-owning kernel (lpx_resident_col.hip; grid / cpw / lds from resident_col_plan), else the row-owning one
int run_resident(lpx_tableau* t, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* stats,
                 int grid, int rpw, size_t lds, int* resume_iter, bool col = false)
{
    const int mcap = t->Rcap;
    { int rc = resident_buffers(t); if (rc) return rc; }
    if (col) { int rc = resident_col_buffers(t, grid); if (rc) return rc; }
    DevState init; std::memset(&init, 0, sizeof(init));
    init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = 2;
    *t->hst = init;
    LPX_HIP_TRY(hipMemcpyAsync(t->st, t->hst, sizeof(DevState), hipMemcpyHostToDevice, t->stream));
    const int chunk = cb ? (o->batch > 0 ? o->batch : 256) : (1 << 30);
    lpx_stats local; std::memset(&local, 0, sizeof(local));
    const double t0 = now_ms();
    int fired = 0, status = LPX_RUNNING;
    for (long long launches = 0; status == LPX_RUNNING; ++launches) {
        if (launches > (long long)o->max_iter + 4) { set_error("resident loop: launch budget exhausted while still running"); return LPX_ITER_LIMIT; }
        // Tableau and basis as of the start of this launch.  A launch that cannot finish normally writes nothing back, but
        // a workgroup that was scheduled late (after the others gave up) may complete a short launch and store its rows:
        // whenever the abort flag is up the host puts this copy back, so the hand-over never sees a half-pivoted tableau.
        LPX_HIP_TRY(hipMemcpyAsync(t->xbasis, t->basis, sizeof(int32_t) * (size_t)(t->R - 1), hipMemcpyDeviceToDevice, t->stream));
        LPX_HIP_TRY(hipMemcpyAsync(t->xT, t->T, sizeof(double) * (size_t)t->R * t->ld, hipMemcpyDeviceToDevice, t->stream));
        if (o->profile) {
            while (t->events.size() < 2) { hipEvent_t e; LPX_HIP_TRY(hipEventCreate(&e)); t->events.push_back(e); }
            LPX_HIP_TRY(hipEventRecord(t->events[0], t->stream));
        }
        if (col)
            LPX_HIP_TRY(launch_resident_primal_col(t->T, t->ld, t->R, t->C, grid, rpw, lds, t->basis, t->trace, t->trace_cap,
                                                   t->st, t->xc, t->xq, t->xgen, o->eps, o->ratio_tol, o->max_iter, chunk, t->stream));
        else
        LPX_HIP_TRY(launch_resident_primal(t->T, t->ld, t->R, t->C, grid, rpw, lds, mcap, t->basis, t->trace, t->trace_cap,
                                           t->st, t->xr, t->xp, t->xgen, o->eps, o->ratio_tol, o->max_iter, chunk, t->stream));
        if (o->profile) LPX_HIP_TRY(hipEventRecord(t->events[1], t->stream));
        local.launches++;
        LPX_HIP_TRY(hipMemcpyAsync(t->hst, t->st, sizeof(DevState), hipMemcpyDeviceToHost, t->stream));
        LPX_HIP_TRY(hipStreamSynchronize(t->stream));
        if (o->profile) {       // HIP events on the library stream around the persistent kernel: its duration
            float ms = 0.f;
            LPX_HIP_TRY(hipEventElapsedTime(&ms, t->events[0], t->events[1]));
            local.update_ms_sum += ms;
            local.update_launches++;
        }
        if (t->hst->pad[1]) {
            // a bounded wait expired: some workgroup was not resident or died; rows in HBM are those of the last
            // completed launch.  Clear the exchange buffers so that no stale generation can ever match.
            resident_buffers_clear(t);
            set_error("resident loop: an exchange wait expired (workgroups not co-resident?)");
            // Put the tableau and the basis of the launch's start back (late workgroups may have stored rows of a
            // pivot the others never made) and hand over to the streaming kernels, which continue from pivot
            // `resume_iter`.
            LPX_HIP_TRY(hipMemcpy(t->T, t->xT, sizeof(double) * (size_t)t->R * t->ld, hipMemcpyDeviceToDevice));
            LPX_HIP_TRY(hipMemcpy(t->basis, t->xbasis, sizeof(int32_t) * (size_t)(t->R - 1), hipMemcpyDeviceToDevice));
            t->resident_off = true;
            *resume_iter = fired;
            return LPX_RESIDENT_RETRY;
        }
        status = t->hst->status;
        const int done = t->hst->iter;
        if (cb && done > fired) {
            const int lo = fired, hi = done < t->trace_cap ? done : t->trace_cap;
            if (hi > lo) {
                std::vector<int32_t> tr(2 * (size_t)(hi - lo));
                LPX_HIP_TRY(hipMemcpy(tr.data(), t->trace + 2 * lo, sizeof(int32_t) * 2 * (hi - lo), hipMemcpyDeviceToHost));
                for (int k = lo; k < hi; ++k) cb(user, k + 1, tr[2 * (k - lo)], tr[2 * (k - lo) + 1]);
            }
        }
        fired = done;
    }
    local.loop_ms = now_ms() - t0;
    local.pivots = t->hst->iter;
    if (stats) { const double h2d = stats->h2d_ms, d2h = stats->d2h_ms; *stats = local; stats->h2d_ms = h2d; stats->d2h_ms = d2h; }
    return status;
}

// Resident group run: the nodes of a batch are solved a few at a time, each resident in the LDS of its own slice
// of the chip (lpx_resident_group.hip).  Launches are `chunk` pivots long; after each one finished nodes leave and
// waiting ones take their place, so the slices stay busy until the batch is done.
struct ResGroupBuf { ResNode* d = nullptr; ResNode* h = nullptr; DevState* hs = nullptr; int cap = 0; hipStream_t stream = nullptr;
                     ParkDesc* pd_d = nullptr; ParkDesc* pd_h = nullptr; };   // pd: descriptors of the snapshot copies (one launch for a whole group)
ResGroupBuf g_resgroup;

// which kernel the last resident_group_plan chose: 0 = rows in LDS (lpx_resident_group), else the workgroup size of the
// register-resident variant (lpx_resident_group_r); read by run_resident_group right after (handles are used from one thread)
static thread_local int tl_plan_nt = 0, tl_plan_rt = 0;     // register-resident kernel: configuration, rows a workgroup may hold

// The register-resident variant (node rows in VGPRs): more nodes per launch when a node is wide enough to need many CUs' LDS.
// Returns the nodes per launch it would give (0 = not applicable) and fills grid / lds / nt.
static int resident_regs_plan(lpx_tableau** ts, int count, int cus, int* grid, size_t* lds, int* nt, int* rt)
{
    static const bool enabled = [] { const char* e = std::getenv("LPX_RESIDENT_REGS"); return !(e && e[0] == '0'); }();
    if (!enabled) return 0;
    int maxC = 2, mmax = 1, mmin = 1 << 30, min_ld = 1 << 30;
    for (int i = 0; i < count; ++i) { maxC = std::max(maxC, ts[i]->C); mmax = std::max(mmax, ts[i]->R - 1); mmin = std::min(mmin, ts[i]->R - 1); min_ld = std::min(min_ld, ts[i]->ld); }
    int rpw_max = 0;
    const int n = resident_regs_shape(maxC, min_ld, mmax, &rpw_max); // the kernel configuration
    if (!n) return 0;
    int g = (mmax + rpw_max - 1) / rpw_max;                 // workgroups per node: every node's rows per workgroup <= rpw_max
    if (g > mmin || g > cus) return 0;
    { const int rpw = (mmax + g - 1) / g; g = (mmax + rpw - 1) / rpw; }          // no idle workgroups for the tallest node
    size_t need = 0;
    for (int i = 0; i < count; ++i) need = std::max(need, resident_regs_lds(ts[i]->R, ts[i]->C, rpw_max, n));
    if (need > resident_regs_lds_budget()) return 0;
    *grid = g; *lds = need; *nt = n; *rt = rpw_max;
    return cus / g;
}

int resident_group_plan(lpx_tableau** ts, int count, int* grid, int* slots, size_t* lds)
{
    tl_plan_nt = 0;
    hipDeviceProp_t prop; int dev = 0;
    static int cus = 0;
    if (!cus) { if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0; cus = prop.multiProcessorCount; }
    const size_t lds_max = 160 * 1024 - 1024;
    // As many nodes per launch as fit: a node takes cus / n workgroups, down to ONE (small node LPs: 240 of them side by side,
    // each in the LDS of one CU).  r01 / early r02 stopped at 8 nodes per launch; a 60-variable 0/1 program went from 7.2 k to
    // 21 k nodes/s when the limit fell (tools/probe_slots.py), node logs and pivot counts unchanged.  LPX_GROUP_SLOTS caps it.
    static const int max_slots = [] { const char* e = std::getenv("LPX_GROUP_SLOTS"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 1 << 20; }();
    // (never more nodes than CUs: a node needs at least one workgroup -- a group of more than 256 nodes divided by zero here before r03)
    for (int n = std::min(std::min(count, max_slots), cus); n >= 1; --n) {
        int g = cus / n;
        size_t need = 0;
        for (int i = 0; i < count; ++i) {
            const int m = ts[i]->R - 1;
            const int gi = g < m ? g : m;
            const size_t b = resident_group_lds(ts[i]->R, ts[i]->C, ts[i]->ld, gi);
            if (b > need) need = b;
        }
        int mmin = 1 << 30;
        for (int i = 0; i < count; ++i) if (ts[i]->R - 1 < mmin) mmin = ts[i]->R - 1;
        if (g > mmin) g = mmin;                       // at most one workgroup per row of the smallest node
        if (g < 1) continue;
        {   // no idle workgroups: the fewest that keep the same rows-per-workgroup for the tallest node
            int mmax = 1;
            for (int i = 0; i < count; ++i) if (ts[i]->R - 1 > mmax) mmax = ts[i]->R - 1;
            const int rpw = (mmax + g - 1) / g;
            g = (mmax + rpw - 1) / rpw;
        }
        need = 0;
        for (int i = 0; i < count; ++i) { const size_t b = resident_group_lds(ts[i]->R, ts[i]->C, ts[i]->ld, g); if (b > need) need = b; }
        if (need <= lds_max) {
            *grid = g; *slots = n; *lds = need;
            // rows in registers instead, when that puts more nodes on the chip at once (and there are enough nodes to use them)
            int rg = 0, rnt = 0, rrt = 0; size_t rlds = 0;
            const int rslots = resident_regs_plan(ts, count, cus, &rg, &rlds, &rnt, &rrt);
            static const bool force_regs = [] { const char* e = std::getenv("LPX_RESIDENT_REGS"); return e && e[0] == '2'; }();   // diagnostic: whenever it applies
            if (rslots >= 1 && (force_regs || (rslots > n && count > n))) { *grid = rg; *slots = std::min(rslots, std::min(count, max_slots)); *lds = rlds; tl_plan_nt = rnt; tl_plan_rt = rrt; }
            return 1;
        }
    }
    {   // nothing fits the LDS form: the register form alone
        int rg = 0, rnt = 0, rrt = 0; size_t rlds = 0;
        const int rslots = resident_regs_plan(ts, count, cus, &rg, &rlds, &rnt, &rrt);
        if (rslots >= 1) { *grid = rg; *slots = std::min(rslots, std::min(count, max_slots)); *lds = rlds; tl_plan_nt = rnt; tl_plan_rt = rrt; return 1; }
    }
    return 0;
}

int run_resident_group(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts, const lpx_run_opts* dopts,
                       int* statuses, lpx_stats* stats, int grid, int slots, size_t lds, lpx_pivot_cb cb, void* user,
                       DevState* resume = nullptr)
{
    ResGroupBuf& g = g_resgroup;
    const int plan_nt = tl_plan_nt, plan_rt = tl_plan_rt;   // 0: rows in LDS; else the configuration of the register-resident kernel
    if (!g.stream) LPX_HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    if (g.cap < count) {
        hipFree(g.d); if (g.h) hipHostFree(g.h); if (g.hs) hipHostFree(g.hs);
        hipFree(g.pd_d); if (g.pd_h) hipHostFree(g.pd_h);
        g.d = nullptr; g.h = nullptr; g.hs = nullptr; g.pd_d = nullptr; g.pd_h = nullptr; g.cap = 0;
        const int c = count + 16;
        LPX_HIP_TRY(hipMalloc((void**)&g.d, sizeof(ResNode) * c));
        LPX_HIP_TRY(hipHostMalloc((void**)&g.h, sizeof(ResNode) * c));
        LPX_HIP_TRY(hipHostMalloc((void**)&g.hs, sizeof(DevState) * c));
        LPX_HIP_TRY(hipMalloc((void**)&g.pd_d, sizeof(ParkDesc) * c));
        LPX_HIP_TRY(hipHostMalloc((void**)&g.pd_h, sizeof(ParkDesc) * c));
        g.cap = c;
    }
    const double t0 = now_ms();
    std::vector<ResNode> node(count);
    // Per-node host work adds up when a group is a whole B&B level (8000 warm-started nodes: 0.24 s of stream waits, state uploads and
    // snapshot copies in front of 0.3 s of kernel): every distinct stream is waited for once, the state records go up in one launch,
    // the snapshots of a launch are one multi-copy launch.
    std::vector<hipStream_t> waited;
    auto wait_once = [&](hipStream_t st) -> int {
        for (hipStream_t w : waited) if (w == st) return 0;
        LPX_HIP_TRY(hipStreamSynchronize(st));
        waited.push_back(st);
        return 0;
    };
    for (int i = 0; i < count; ++i) {
        lpx_tableau* t = ts[i];
        { int rc = wait_once(t->stream); if (rc) return rc; }          // node assembly ran on the node's own stream
        if (!t->xr) {
            { int rc = resident_buffers(t); if (rc) return rc; }
            LPX_HIP_TRY(hipStreamSynchronize(t->stream));
        }
        const lpx_run_opts* o = dual[i] ? dopts : popts;
        ResNode& n = node[i];
        n.T = t->T; n.ld = t->ld; n.R = t->R; n.C = t->C; n.basis = t->basis; n.trace = t->trace; n.trace_cap = t->trace_cap;
        n.st = t->st; n.xr = t->xr; n.xp = t->xp; n.xgen = t->xgen; n.mcap = t->Rcap; n.dual = dual[i] ? 1 : 0;
        n.eps = o->eps; n.tol_fdf = o->ratio_tol; n.tol_dual = o->ratio_tol; n.tol_primal = dual[i] ? o->eps : o->ratio_tol;
        n.max_iter = o->max_iter; n.fdf_guard = o->fdf_guard; n.cleanup = o->cleanup;
        DevState init; std::memset(&init, 0, sizeof(init));
        init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = dual[i] ? 0 : 2;
        g.hs[i] = init;
        n.st_host = &g.hs[i];
    }
    // every node's initial state record: one launch reading the pinned array (node i <-> g.hs[i])
    std::memcpy(g.h, node.data(), sizeof(ResNode) * (size_t)count);
    LPX_HIP_TRY(hipMemcpyAsync(g.d, g.h, sizeof(ResNode) * (size_t)count, hipMemcpyHostToDevice, g.stream));
    LPX_HIP_TRY(launch_resnode_states_scatter(g.d, g.hs, count, g.stream));
    LPX_HIP_TRY(hipStreamSynchronize(g.stream));                        // g.h / g.d are rewritten per launch below
    // launch length: long enough to hide the launch + reload (~30 us), short enough that a node finishing inside a
    // launch does not leave its slice idle for long
    const lpx_run_opts* o0 = dual[0] ? dopts : popts;
    static const int chunk_env = [] { const char* e = std::getenv("LPX_GROUP_CHUNK"); return e ? std::atoi(e) : 0; }();   // diagnostic
    // A group larger than the chip holds at a time goes out as ONE launch all the same (r03): the hardware hands workgroups to compute
    // units in launch order, so the workgroups of node `slots` + k start as those of an earlier node leave -- a finished node's
    // successor starts at once instead of at the next launch boundary, where the chip used to wait for the host (9 % of the cold
    // config-4 search) and for the slowest node of the launch (6 %).  The earliest incomplete node is first in line for every unit
    // that frees up, so it always completes its set; its early workgroups poll meanwhile (bounded waits of ~0.6 s against node
    // run times of milliseconds).  LPX_GROUP_WALK=0: launches of `slots` nodes and 96 pivots, refilled by the host in between.
    static const bool walk_env = [] { const char* e = std::getenv("LPX_GROUP_WALK"); return !(e && e[0] == '0'); }();
    const bool walk = walk_env && cb == nullptr && count > slots;
    const int chunk = cb ? (o0->batch > 0 ? o0->batch : 256) : (walk ? (1 << 20) : (count > slots ? (chunk_env > 0 ? chunk_env : 96) : 1024));
    std::vector<int> live(count);
    for (int i = 0; i < count; ++i) live[i] = i;
    std::vector<int> fired(count, 0);
    std::vector<DevState> before(count);
    std::vector<char> snapped(count, 0);
    // What an aborted launch goes back to: with a pivot callback the state at the START OF THAT LAUNCH (the callbacks of the
    // earlier launches have fired), snapshot per launch; without one (B&B batches) the node's state at its FIRST launch --
    // one snapshot per node instead of one per launch (a node of config 4 takes eight launches), the rare restart repeats
    // the node's pivots on the streaming kernels and ends in the same tableau.
    const bool snap_each_launch = cb != nullptr;
    long long launches = 0;
    while (!live.empty()) {
        const int n = walk ? (int)std::min<size_t>(live.size(), 65535) : ((int)live.size() < slots ? (int)live.size() : slots);   // gridDim.y <= 65535
        int nsnap = 0; size_t maxd = 2;
        for (int k = 0; k < n; ++k) {
            g.h[k] = node[live[k]];
            lpx_tableau* t = ts[live[k]];
            if (snap_each_launch || !snapped[live[k]]) {
                before[live[k]] = g.hs[live[k]];
                ParkDesc& d = g.pd_h[nsnap++];
                d.srcT = t->T; d.dstT = t->xT; d.srcB = t->basis; d.dstB = t->xbasis;
                d.doubles = (size_t)t->R * t->ld; d.m = t->R - 1; d.pad = 0;
                maxd = std::max(maxd, d.doubles);
                snapped[live[k]] = 1;
            }
        }
        if (nsnap > 0) {                        // the snapshots of this launch: one multi-copy launch (lpx_park_many)
            LPX_HIP_TRY(hipMemcpyAsync(g.pd_d, g.pd_h, sizeof(ParkDesc) * (size_t)nsnap, hipMemcpyHostToDevice, g.stream));
            const int bpn = (int)std::min<size_t>(256, std::max<size_t>(1, maxd / 2 / 256 / 4));
            LPX_HIP_TRY(launch_park_many(g.pd_d, nsnap, bpn, g.stream));
        }
        // the kernel writes each node's new state into the pinned mirror (ResNode::st_host): nothing is copied back
        LPX_HIP_TRY(hipMemcpyAsync(g.d, g.h, sizeof(ResNode) * n, hipMemcpyHostToDevice, g.stream));
        if (plan_nt) LPX_HIP_TRY(launch_resident_regs(g.d, n, grid, plan_nt, plan_rt, lds, chunk, g.stream));
        else LPX_HIP_TRY(launch_resident_group(g.d, n, grid, lds, chunk, g.stream));
        LPX_HIP_TRY(hipStreamSynchronize(g.stream));
        bool aborted = false;
        for (int k = 0; k < n; ++k) if (g.hs[live[k]].pad[1]) aborted = true;
        if (aborted) {
            for (int k = 0; k < n; ++k) resident_buffers_clear(ts[live[k]]);
            set_error("resident group loop: an exchange wait expired (workgroups not co-resident?)");
            // A node whose launch aborted goes back to the state of the launch's start (tableau, basis, counters: a late
            // workgroup may have stored rows of a pivot the others never made); the other nodes of the launch finished it
            // normally and keep what they wrote.  The caller finishes every unfinished node on the streaming kernels.
            for (int k = 0; k < n; ++k) {
                if (!g.hs[live[k]].pad[1]) continue;
                lpx_tableau* t = ts[live[k]];
                LPX_HIP_TRY(hipMemcpy(t->T, t->xT, sizeof(double) * (size_t)t->R * t->ld, hipMemcpyDeviceToDevice));
                LPX_HIP_TRY(hipMemcpy(t->basis, t->xbasis, sizeof(int32_t) * (size_t)(t->R - 1), hipMemcpyDeviceToDevice));
                g.hs[live[k]] = before[live[k]];
                g.hs[live[k]].pad[1] = 0;
                LPX_HIP_TRY(hipMemcpy(t->st, &g.hs[live[k]], sizeof(DevState), hipMemcpyHostToDevice));
            }
            for (int i = 0; i < count; ++i) statuses[i] = g.hs[i].status;      // LPX_RUNNING marks the unfinished ones
            if (resume) std::memcpy(resume, g.hs, sizeof(DevState) * count);
            return LPX_RESIDENT_RETRY;
        }
        ++launches;
        if (launches > 4LL * count * ((long long)popts->max_iter + dopts->max_iter + dopts->fdf_guard) / chunk + 64) {
            set_error("resident group loop: launch budget exhausted while still running"); return LPX_ITER_LIMIT; }
        std::vector<int> next;
        for (int k = 0; k < n; ++k) {
            const int i = live[k];
            const DevState& s = g.hs[i];
            if (cb && s.iter > fired[i]) {
                lpx_tableau* t = ts[i];
                const int lo = fired[i], hi = s.iter < t->trace_cap ? s.iter : t->trace_cap;
                if (hi > lo) {
                    std::vector<int32_t> tr(2 * (size_t)(hi - lo));
                    LPX_HIP_TRY(hipMemcpy(tr.data(), t->trace + 2 * lo, sizeof(int32_t) * 2 * (hi - lo), hipMemcpyDeviceToHost));
                    for (int z = lo; z < hi; ++z) cb(user, z + 1, tr[2 * (z - lo)], tr[2 * (z - lo) + 1]);
                }
                fired[i] = s.iter;
            }
            if (s.status == LPX_RUNNING) next.push_back(i);
        }
        for (size_t k = (size_t)n; k < live.size(); ++k) next.push_back(live[k]);
        live.swap(next);
    }
    const double ms = now_ms() - t0;
    for (int i = 0; i < count; ++i) {
        const DevState& s = g.hs[i];
        *ts[i]->hst = s;
        statuses[i] = s.status;
        if (stats) {
            std::memset(&stats[i], 0, sizeof(lpx_stats));
            stats[i].pivots = s.iter; stats[i].fdf_pivots = s.fdf_count;
            stats[i].cleanup_pivots = dual[i] ? s.primal_count : 0;
            stats[i].loop_ms = ms / (double)count;
            stats[i].launches = launches;
        }
    }
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Batched group run (K9): all tableaux of a group advance one pivot per launch pair (blockIdx.y = node),
// so many small node LPs fill the chip from ONE stream instead of competing for a few hardware queues.
// ---------------------------------------------------------------------------------------------------
namespace {

struct GroupBuf {
    SelParams* d = nullptr; SelParams* h = nullptr; DevState* hs = nullptr; int cap = 0;
    hipStream_t stream = nullptr;
    hipGraphExec_t gexec = nullptr; std::string gkey;
};
GroupBuf g_groups[2];        // [primal, dual]; lpx handles are used from one thread per process

int group_reserve(GroupBuf& g, int count)
{
    if (!g.stream) LPX_HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    if (count <= g.cap) return 0;
    if (g.gexec) { hipGraphExecDestroy(g.gexec); g.gexec = nullptr; g.gkey.clear(); }
    hipFree(g.d); if (g.h) hipHostFree(g.h); if (g.hs) hipHostFree(g.hs);
    g.d = nullptr; g.h = nullptr; g.hs = nullptr; g.cap = 0;
    const int c = count + 16;
    LPX_HIP_TRY(hipMalloc((void**)&g.d, sizeof(SelParams) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.h, sizeof(SelParams) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.hs, sizeof(DevState) * c));
    g.cap = c;
    return 0;
}

struct GroupRun {
    GroupBuf* g = nullptr; std::vector<int> idx; int dual = 0; int batch = 64; long long budget = 0, enq = 0;
    int max_nblk = 1, max_blocks = 1, maxR = 2, maxC = 2; bool done = true;
    int min_active = 0; bool suspended_exit = false;     // stop (without exhausting the budget) once this few nodes are still running
};

int group_begin(GroupRun& r, lpx_tableau** ts, const lpx_run_opts* o, const DevState* inits = nullptr)
{
    GroupBuf& g = *r.g;
    const int K = (int)r.idx.size();
    int rc = group_reserve(g, K); if (rc) return rc;
    r.batch = o->batch > 0 ? o->batch : 64;
    r.budget = r.dual ? (long long)o->fdf_guard + 2LL * o->max_iter + 8 : (long long)o->max_iter + 2;
    r.max_nblk = 1; r.max_blocks = 1; r.maxR = 2; r.maxC = 2;
    for (int k = 0; k < K; ++k) {
        lpx_tableau* t = ts[r.idx[k]];
        LPX_HIP_TRY(hipStreamSynchronize(t->stream));               // node assembly ran on the node's own stream
        SelParams p = base_params(t, o, r.dual ? MODE_DUAL : MODE_PRIMAL);
        if (!r.dual && !p.us) { set_error("batched primal run needs the multi-workgroup select"); return LPX_EINVAL; }
        g.h[k] = p;
        if (p.nblk > r.max_nblk) r.max_nblk = p.nblk;
        const int ub = update_blocks(t->ld, t->Rcap);
        if (ub > r.max_blocks) r.max_blocks = ub;
        if (t->Rcap > r.maxR) r.maxR = t->Rcap;
        if (t->Ccap > r.maxC) r.maxC = t->Ccap;
        DevState init; std::memset(&init, 0, sizeof(init));
        init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = r.dual ? 0 : 2;
        if (inits) {                                // continue where another path stopped (pivot count, phase, counters)
            const DevState& s0 = inits[r.idx[k]];
            init.iter = s0.iter; init.phase = r.dual ? s0.phase : 2;
            init.fdf_count = s0.fdf_count; init.dual_iter = s0.dual_iter; init.primal_count = s0.primal_count;
        }
        g.hs[k] = init;
    }
    LPX_HIP_TRY(hipMemcpyAsync(g.d, g.h, sizeof(SelParams) * K, hipMemcpyHostToDevice, g.stream));
    LPX_HIP_TRY(launch_states_scatter(g.d, g.hs, K, g.stream));      // every node's initial state record, one launch
    if (!r.dual) LPX_HIP_TRY(launch_group_init(g.d, K, g.stream));
    else LPX_HIP_TRY(launch_group_rhs_init(g.d, K, g.stream));
    // graph of `batch` iterations, keyed by everything baked into the launches
    char keybuf[160];
    std::snprintf(keybuf, sizeof(keybuf), "%p/%d/%d/%d/%d/%d/%d/%d", (void*)g.d, K, r.dual, r.max_nblk, r.max_blocks, r.batch, r.maxR, r.maxC);
    if (o->use_graph && g.gkey != keybuf) {
        if (g.gexec) { hipGraphExecDestroy(g.gexec); g.gexec = nullptr; }
        LPX_HIP_TRY(hipStreamSynchronize(g.stream));
        hipGraph_t graph = nullptr;
        LPX_HIP_TRY(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < r.batch; ++i) {
            hipError_t e = launch_group_iter(g.d, K, r.dual, r.max_nblk, r.max_blocks, g.stream, r.maxR, r.maxC);
            if (e != hipSuccess) { hipStreamEndCapture(g.stream, &graph); if (graph) hipGraphDestroy(graph); set_error("group capture failed"); return LPX_EDEVICE; }
        }
        LPX_HIP_TRY(hipStreamEndCapture(g.stream, &graph));
        hipError_t e = hipGraphInstantiate(&g.gexec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (e != hipSuccess) { g.gexec = nullptr; set_error("group graph instantiate failed"); return LPX_EDEVICE; }
        g.gkey = keybuf;
    }
    r.enq = 0; r.done = false;
    return 0;
}

int group_submit(GroupRun& r, lpx_tableau** ts, const lpx_run_opts* o)
{
    GroupBuf& g = *r.g;
    const int K = (int)r.idx.size();
    if (o->use_graph && g.gexec) LPX_HIP_TRY(hipGraphLaunch(g.gexec, g.stream));
    else for (int i = 0; i < r.batch; ++i) LPX_HIP_TRY(launch_group_iter(g.d, K, r.dual, r.max_nblk, r.max_blocks, g.stream, r.maxR, r.maxC));
    r.enq += r.batch;
    (void)ts;
    LPX_HIP_TRY(launch_states_gather(g.d, g.hs, K, g.stream));       // every node's state record into the pinned array, one launch
    return 0;
}

int group_complete(GroupRun& r)
{
    GroupBuf& g = *r.g;
    LPX_HIP_TRY(hipStreamSynchronize(g.stream));
    int running = 0;
    for (size_t k = 0; k < r.idx.size(); ++k) if (g.hs[k].status == LPX_RUNNING) ++running;
    r.done = running == 0 || r.enq >= r.budget;
    if (!r.done && r.min_active > 0 && running <= r.min_active) { r.done = true; r.suspended_exit = true; }
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// Fused group run (K4g, lpx_group_fused): ONE launch per step for the whole group -- update(k) of every live node out of place
// beside select(k+1) of every live node -- and a live list instead of early-exit workgroups: between polls the host drops the
// finished nodes from the grid.  Launches are eager (a launch costs the host ~5 us against a step of tens of microseconds on
// the device, and the grid changes from poll to poll).  Every node needs its second tableau buffer (fused_buffers); a group in
// which one does not get it runs on the two-launch kernels below.  LPX_GROUP_FUSED=0: never (diagnostic).
// ---------------------------------------------------------------------------------------------------
namespace {

struct FusedGroupBuf {
    FusedParams* d = nullptr; FusedParams* h = nullptr;     // parameter records: device / pinned
    DevState* hs = nullptr;                                  // pinned: initial states in, latest records out
    DevState* ds = nullptr;                                  // device copy of the initial states
    int* live_d = nullptr; int* live_h = nullptr;            // live list (two halves: launches of window w read half w & 1)
    int* comp_d = nullptr; int* comp_h = nullptr;            // device-side compaction record (lpx_kernels.hip FG_COMP_*) and its pinned staging
    int* fresh_d = nullptr; int* fresh_h = nullptr;
    int* cur_h = nullptr;                                    // pinned: index of every node's latest record
    int cap = 0;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> ev;
};

static constexpr int kAsyncSlots = 4;           // rolling batches a host may keep in flight (lpx_multi_run_begin / _end)
extern FusedGroupBuf g_fgroups[kAsyncSlots + 1];
int fused_group_reserve(FusedGroupBuf& g, int count)
{
    static const bool one_stream = [] { const char* e = std::getenv("LPX_ROLL_ONE_STREAM"); return e && e[0] == '1'; }();   // experiment
    if (!g.stream && one_stream && &g >= &g_fgroups[0] && &g < &g_fgroups[kAsyncSlots])
        for (int k = 0; k < kAsyncSlots && !g.stream; ++k) if (g_fgroups[k].stream) g.stream = g_fgroups[k].stream;
    if (!g.stream) {
        // LOWEST priority: a window is a dozen chip-filling launches in a row; the small launches the host needs answered while the
        // other batch pivots (solution read-back, parking, child assembly: other streams, default priority) must get their
        // workgroups in as slots free up instead of queueing behind the window (measured: they waited 1-2 ms each without this)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = 0; }
        if (hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            LPX_HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        }
    }
    if (count <= g.cap) return 0;
    hipFree(g.d); hipFree(g.ds); hipFree(g.live_d); hipFree(g.fresh_d); hipFree(g.comp_d);
    if (g.comp_h) hipHostFree(g.comp_h);
    if (g.h) hipHostFree(g.h); if (g.hs) hipHostFree(g.hs); if (g.live_h) hipHostFree(g.live_h);
    if (g.fresh_h) hipHostFree(g.fresh_h); if (g.cur_h) hipHostFree(g.cur_h);
    { hipStream_t st = g.stream; std::vector<hipEvent_t> ev = std::move(g.ev); g = FusedGroupBuf{}; g.stream = st; g.ev = std::move(ev); }
    const int c = count + 16;
    LPX_HIP_TRY(hipMalloc((void**)&g.d, sizeof(FusedParams) * c));
    LPX_HIP_TRY(hipMalloc((void**)&g.ds, sizeof(DevState) * c));
    LPX_HIP_TRY(hipMalloc((void**)&g.live_d, sizeof(int) * 2 * c));
    LPX_HIP_TRY(hipMalloc((void**)&g.fresh_d, sizeof(int) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.h, sizeof(FusedParams) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.hs, sizeof(DevState) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.live_h, sizeof(int) * 2 * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.fresh_h, sizeof(int) * c));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.cur_h, sizeof(int) * c));
    LPX_HIP_TRY(hipMalloc((void**)&g.comp_d, sizeof(int) * group_fused_comp_ints(c)));
    LPX_HIP_TRY(hipHostMalloc((void**)&g.comp_h, sizeof(int) * 2 * (32 + c)));       // two windows' worth of {counts, list of parity 0}
    LPX_HIP_TRY(hipMemsetAsync(g.comp_d, 0, sizeof(int) * group_fused_comp_ints(c), g.stream));
    g.cap = c;
    return 0;
}

FusedGroupBuf g_fgroups[kAsyncSlots + 1];   // [0 .. kAsyncSlots): the asynchronous batches (lpx_multi_run_begin / _end), [kAsyncSlots]: the synchronous runs

// parameter records, initial states and the init launch of a group; returns LPX_RESIDENT_RETRY when the group cannot take the
// fused path (nothing has been touched then)
int fused_prepare(FusedGroupBuf& g, lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts, const lpx_run_opts* dopts,
                  const DevState* inits, int* per_node_out, int* batch_out, long long* budget_out)
{
    static const bool enabled = [] { const char* e = std::getenv("LPX_GROUP_FUSED"); return !(e && e[0] == '0'); }();
    if (!enabled || count < 1) return LPX_RESIDENT_RETRY;
    for (int i = 0; i < count; ++i) if (ts[i]->fused_off) return LPX_RESIDENT_RETRY;
    for (int i = 0; i < count; ++i) if (!fused_buffers(ts[i])) return LPX_RESIDENT_RETRY;
    { int rc = fused_group_reserve(g, count); if (rc) return rc; }
    int per_node = 1, nfresh = 0, batch = 64;
    long long budget = 0;
    for (int k = 0; k < count; ++k) {
        lpx_tableau* t = ts[k];
        LPX_HIP_TRY(hipStreamSynchronize(t->stream));               // node assembly ran on the node's own stream
        const lpx_run_opts* o = dual[k] ? dopts : popts;
        FusedParams f; std::memset(&f, 0, sizeof(f));
        f.P = base_params(t, o, dual[k] ? MODE_DUAL : MODE_PRIMAL);
        f.T1 = t->fT; f.prow1 = t->fprow; f.rhs1 = t->frhs; f.rec = t->frec;
        const bool cont = t->fsuspended;                            // continues a fused run: its records are in place
        f.par = cont ? (t->frec_cur & 1) : 0;
        g.h[k] = f;
        DevState init; std::memset(&init, 0, sizeof(init));
        init.status = LPX_RUNNING; init.r = -1; init.q = -1; init.qn = -1; init.phase = dual[k] ? 0 : 2;
        if (inits) {                                                // continue where another path stopped (pivot count, phase, counters)
            const DevState& s0 = inits[k];
            init.iter = s0.iter; init.phase = dual[k] ? s0.phase : 2;
            init.fdf_count = s0.fdf_count; init.dual_iter = s0.dual_iter; init.primal_count = s0.primal_count;
        }
        g.hs[k] = init;
        if (!cont) g.fresh_h[nfresh++] = k;
        t->fsuspended = false;
        per_node = std::max(per_node, group_fused_blocks(t->ld, t->Rcap));
        batch = o->batch > 0 ? o->batch : 64;
        budget = std::max(budget, dual[k] ? (long long)o->fdf_guard + 2LL * o->max_iter + 12 : (long long)o->max_iter + 6);
    }
    batch = (batch + 1) & ~1;           // launches alternate between the two record indices: a window ends where it started
    LPX_HIP_TRY(hipMemcpyAsync(g.d, g.h, sizeof(FusedParams) * count, hipMemcpyHostToDevice, g.stream));
    if (nfresh > 0) {
        LPX_HIP_TRY(hipMemcpyAsync(g.ds, g.hs, sizeof(DevState) * count, hipMemcpyHostToDevice, g.stream));
        LPX_HIP_TRY(hipMemcpyAsync(g.fresh_d, g.fresh_h, sizeof(int) * nfresh, hipMemcpyHostToDevice, g.stream));
        LPX_HIP_TRY(launch_group_fused_init(g.d, g.fresh_d, nfresh, g.ds, g.stream));
    }
    *per_node_out = per_node; *batch_out = batch; *budget_out = budget;
    return 0;
}

// what a run leaves on its handles and reports: latest records in g.hs / g.cur_h
void fused_finish(FusedGroupBuf& g, lpx_tableau** ts, const int* dual, int count, bool unfinished_is_suspended, double ms, long long enq,
                  int* statuses, lpx_stats* stats, double prof_ms, long long prof_n)
{
    for (int k = 0; k < count; ++k) {
        lpx_tableau* t = ts[k];
        const DevState& s = g.hs[k];
        *t->hst = s;
        const bool running = s.status == LPX_RUNNING;
        statuses[k] = running ? (unfinished_is_suspended ? LPX_RUNNING : LPX_ITER_LIMIT) : s.status;
        t->suspended = statuses[k] == LPX_RUNNING;
        t->fsuspended = t->suspended;
        t->frec_cur = g.cur_h[k];
        // a finished node whose last pivot landed in the second buffer: the buffers trade places (every consumer -- solution
        // read-back, parking, child assembly, download -- goes through t->T); an unfinished one keeps its pending pivot where it is
        if (!running && s.pad[3] == 1) { std::swap(t->T, t->fT); drop_graph(t); }
        if (stats) {
            const double h2d = stats[k].h2d_ms, d2h = stats[k].d2h_ms;      // one-shot entry points keep their transfer times here
            std::memset(&stats[k], 0, sizeof(lpx_stats));
            stats[k].h2d_ms = h2d; stats[k].d2h_ms = d2h;
            stats[k].pivots = s.iter; stats[k].fdf_pivots = s.fdf_count;
            stats[k].cleanup_pivots = dual[k] ? s.primal_count : 0;
            stats[k].loop_ms = ms / (double)count;
            stats[k].launches = enq / (long long)count + 1;
            if (k == 0) { stats[k].update_ms_sum = prof_ms; stats[k].update_launches = prof_n; }   // group-level figures
        }
    }
}

int multi_run_fused(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts, const lpx_run_opts* dopts,
                    int* statuses, lpx_stats* stats, const DevState* inits, int min_active)
{
    FusedGroupBuf& g = g_fgroups[kAsyncSlots];
    const double t0 = now_ms();
    int per_node = 1, batch = 64; long long budget = 0;
    { const int rc = fused_prepare(g, ts, dual, count, popts, dopts, inits, &per_node, &batch, &budget); if (rc) return rc; }
    std::vector<int> live(count);
    for (int k = 0; k < count; ++k) live[k] = k;
    long long enq = 0; int window = 0; bool suspended_exit = false;
    const bool profile = popts->profile || dopts->profile;      // every launch bracketed by HIP events bound to the dispatch
    double prof_ms = 0.0; long long prof_n = 0;
    if (profile) while ((int)g.ev.size() < 2 * batch) { hipEvent_t e; LPX_HIP_TRY(hipEventCreate(&e)); g.ev.push_back(e); }
    while (!live.empty() && enq < budget) {
        // this window's live list (its own half of the buffer: the previous window's launches may still be reading theirs -- they
        // are not, the poll below waits, but the copy stays safe if the loop ever runs ahead)
        int* lh = g.live_h + (window & 1) * g.cap; int* ld_ = g.live_d + (window & 1) * g.cap;
        size_t live_bytes = 0;
        for (size_t k = 0; k < live.size(); ++k) { lh[k] = live[k]; const lpx_tableau* t = ts[live[k]]; live_bytes += sizeof(double) * (size_t)t->R * t->ld; }
        LPX_HIP_TRY(hipMemcpyAsync(ld_, lh, sizeof(int) * live.size(), hipMemcpyHostToDevice, g.stream));
        {   // the device's own live list starts the window equal to the host's (parity 0: windows are even)
            const int hdr = group_fused_comp_hdr();
            int* ch = g.comp_h + (window & 1) * (32 + g.cap);
            std::memset(ch, 0, sizeof(int) * hdr);
            ch[0] = (int)live.size();
            for (size_t k = 0; k < live.size(); ++k) ch[hdr + k] = live[k];
            LPX_HIP_TRY(hipMemcpyAsync(g.comp_d, ch, sizeof(int) * (hdr + live.size()), hipMemcpyHostToDevice, g.stream));
        }
        for (int i = 0; i < batch; ++i)
            LPX_HIP_TRY(launch_group_fused(g.d, ld_, (int)live.size(), per_node, (int)((enq + i) & 1), live_bytes, g.stream, g.comp_d, g.cap,
                                           profile ? g.ev[2 * i] : nullptr, profile ? g.ev[2 * i + 1] : nullptr));
        enq += batch;
        LPX_HIP_TRY(launch_group_fused_gather(g.d, count, g.hs, g.cur_h, g.stream));
        LPX_HIP_TRY(hipStreamSynchronize(g.stream));
        if (profile) {
            // launches that applied a pivot of at least one node: those up to the largest pivot count of the window's live nodes
            int full = 0;
            for (int k : live) full = std::max(full, g.hs[k].iter);
            const long long first = enq - batch;                   // launch l applies pivot l (the first launch of a run applies none)
            for (int i = 0; i < batch; ++i) {
                if (first + i < 1 || first + i > full) continue;
                float msf = 0.f;
                LPX_HIP_TRY(hipEventElapsedTime(&msf, g.ev[2 * i], g.ev[2 * i + 1]));
                prof_ms += msf; ++prof_n;
            }
        }
        ++window;
        std::vector<int> next;
        for (int k : live) if (g.hs[k].status == LPX_RUNNING) next.push_back(k);
        live.swap(next);
        if (!live.empty() && min_active > 0 && (int)live.size() <= min_active) { suspended_exit = true; break; }
    }
    fused_finish(g, ts, dual, count, suspended_exit, now_ms() - t0, enq, statuses, stats, prof_ms, prof_n);
    return 0;
}

// ---- the same in two halves: one window of `steps` pivots of every run of a batch, enqueued and collected separately, so that the
//      host can work on one batch (read-back, parking, assembly of the next nodes) while the other one pivots ----
struct FusedAsync { bool active = false; std::vector<lpx_tableau*> ts; std::vector<int> dual; double t0 = 0; long long enq = 0; hipEvent_t done = nullptr; };
FusedAsync g_fasync[kAsyncSlots];

}  // namespace

extern "C" {

int lpx_multi_run_begin(int slot, lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts, const lpx_run_opts* dopts, int steps)
{
    if (slot < 0 || slot >= kAsyncSlots || !ts || !dual || count < 1 || steps < 1) { set_error("lpx_multi_run_begin: bad argument"); return LPX_EINVAL; }
    FusedAsync& a = g_fasync[slot];
    if (a.active) { set_error("lpx_multi_run_begin: this slot has a batch in flight (lpx_multi_run_end first)"); return LPX_EINVAL; }
    lpx_run_opts pd, dd;
    if (!popts) { lpx_default_opts(&pd, 0); popts = &pd; }
    if (!dopts) { lpx_default_opts(&dd, 1); dopts = &dd; }
    for (int i = 0; i < count; ++i) {
        if (!ts[i] || ts[i]->R < 2) { set_error("lpx_multi_run_begin: null or empty tableau"); return LPX_EINVAL; }
        if (ts[i]->suspended2) { set_error("lpx_multi_run_begin: a run suspended on the two-launch kernels cannot continue here"); return LPX_EINVAL; }
    }
    if (popts->profile || dopts->profile) return 1;
    FusedGroupBuf& g = g_fgroups[slot];
    int per_node = 1, batch = 64; long long budget = 0;
    const double t0 = now_ms();
    {
        const int rc = fused_prepare(g, ts, dual, count, popts, dopts, nullptr, &per_node, &batch, &budget);
        if (rc == LPX_RESIDENT_RETRY) return 1;                 // not available for this batch: the caller takes lpx_multi_run_some
        if (rc) return rc;
    }
    steps = (steps + 1) & ~1;
    size_t live_bytes = 0;
    for (int k = 0; k < count; ++k) { g.live_h[k] = k; live_bytes += sizeof(double) * (size_t)ts[k]->R * ts[k]->ld; }
    LPX_HIP_TRY(hipMemcpyAsync(g.live_d, g.live_h, sizeof(int) * count, hipMemcpyHostToDevice, g.stream));
    { const int hdr = group_fused_comp_hdr();
      std::memset(g.comp_h, 0, sizeof(int) * hdr);
      g.comp_h[0] = count;
      for (int k = 0; k < count; ++k) g.comp_h[hdr + k] = k;
      LPX_HIP_TRY(hipMemcpyAsync(g.comp_d, g.comp_h, sizeof(int) * (hdr + count), hipMemcpyHostToDevice, g.stream)); }
    for (int i = 0; i < steps; ++i) LPX_HIP_TRY(launch_group_fused(g.d, g.live_d, count, per_node, i & 1, live_bytes, g.stream, g.comp_d, g.cap));
    LPX_HIP_TRY(launch_group_fused_gather(g.d, count, g.hs, g.cur_h, g.stream));
    if (!a.done) LPX_HIP_TRY(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
    LPX_HIP_TRY(hipEventRecord(a.done, g.stream));              // the two slots may share a stream: wait for THIS window, not for the stream
    a.active = true; a.ts.assign(ts, ts + count); a.dual.assign(dual, dual + count); a.t0 = t0; a.enq = steps;
    return 0;
}

int lpx_multi_run_end(int slot, int* statuses, lpx_stats* stats)
{
    if (slot < 0 || slot >= kAsyncSlots || !statuses) { set_error("lpx_multi_run_end: bad argument"); return LPX_EINVAL; }
    FusedAsync& a = g_fasync[slot];
    if (!a.active) { set_error("lpx_multi_run_end: no batch in flight in this slot"); return LPX_EINVAL; }
    FusedGroupBuf& g = g_fgroups[slot];
    a.active = false;
    LPX_HIP_TRY(hipEventSynchronize(a.done));
    fused_finish(g, a.ts.data(), a.dual.data(), (int)a.ts.size(), true, now_ms() - a.t0, a.enq, statuses, stats, 0.0, 0);
    return 0;
}

}  // extern "C"

namespace {
}  // namespace

static int multi_run_batched(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts,
                             const lpx_run_opts* dopts, int* statuses, lpx_stats* stats, const DevState* inits = nullptr, int min_active = 0)
{
    {   // one launch per step when every node has its second buffer (and none is in the middle of a two-launch run)
        bool any_two_launch = false;
        for (int i = 0; i < count; ++i) if (ts[i]->suspended2) any_two_launch = true;
        if (!any_two_launch) {
            const int rc = multi_run_fused(ts, dual, count, popts, dopts, statuses, stats, inits, min_active);
            if (rc != LPX_RESIDENT_RETRY) return rc;
        }
    }
    GroupRun runs[2];
    for (int w = 0; w < 2; ++w) { runs[w].g = &g_groups[w]; runs[w].dual = w; runs[w].min_active = min_active; }
    for (int i = 0; i < count; ++i) runs[dual[i] ? 1 : 0].idx.push_back(i);
    const double t0 = now_ms();
    for (int w = 0; w < 2; ++w) if (!runs[w].idx.empty()) { int rc = group_begin(runs[w], ts, w ? dopts : popts, inits); if (rc) return rc; }
    for (;;) {
        bool any = false;
        for (int w = 0; w < 2; ++w) if (!runs[w].done) { int rc = group_submit(runs[w], ts, w ? dopts : popts); if (rc) return rc; any = true; }
        if (!any) break;
        for (int w = 0; w < 2; ++w) if (!runs[w].idx.empty() && runs[w].enq > 0 && !runs[w].done) { int rc = group_complete(runs[w]); if (rc) return rc; }
    }
    const double ms = now_ms() - t0;
    for (int w = 0; w < 2; ++w) {
        GroupRun& r = runs[w];
        for (size_t k = 0; k < r.idx.size(); ++k) {
            const DevState& s = r.g->hs[k];
            const int i = r.idx[k];
            *ts[i]->hst = s;
            statuses[i] = s.status == LPX_RUNNING ? (r.suspended_exit ? LPX_RUNNING : LPX_ITER_LIMIT) : s.status;
            ts[i]->suspended = statuses[i] == LPX_RUNNING;          // continues from *hst in the next lpx_multi_run_some
            ts[i]->suspended2 = ts[i]->suspended;
            if (stats) {
                std::memset(&stats[i], 0, sizeof(lpx_stats));
                stats[i].pivots = s.iter; stats[i].fdf_pivots = s.fdf_count;
                stats[i].cleanup_pivots = w ? s.primal_count : 0;
                stats[i].loop_ms = ms / (double)count;
                stats[i].launches = 2 * r.enq / (long long)(r.idx.size() ? r.idx.size() : 1);
            }
        }
    }
    return 0;
}

extern "C" {

int lpx_primal_run(lpx_tableau* t, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    if (!t) { set_error("lpx_primal_run: null tableau"); return LPX_EINVAL; }
    lpx_run_opts d; if (!o) { lpx_default_opts(&d, 0); o = &d; }
    if (t->R < 2) { set_error("lpx_primal_run: tableau needs at least one constraint row"); return LPX_EINVAL; }
    // Tableau small enough to live on chip: persistent workgroups, no per-pivot HBM traffic (lpx_resident.hip).
    static const bool res_env = [] { const char* e = std::getenv("LPX_RESIDENT"); return !(e && e[0] == '0'); }();
    int resume = 0;
    if (o->resident > 0 || (o->resident == 0 && res_env && !o->profile && (o->batch == 0 || o->batch >= 32))) {
        int grid = 0, rpw = 0; size_t lds = 0;
        // LPX_RESIDENT_COL=1: the column-owning variant (lpx_resident_col.hip) when a workgroup's columns fit its LDS.  It was
        // built to save one of the two cross-CU exchanges per pivot and is bit-identical, but measures the same 7.9-8.0 us per
        // pivot on config 2 as the row-owning kernel, so the row-owning one -- which fits more shapes -- stays the default.
        static const bool col_env = [] { const char* e = std::getenv("LPX_RESIDENT_COL"); return e && e[0] == '1'; }();
        bool col = false;
        if (!t->resident_off && col_env && resident_col_plan(t->R, t->C, &grid, &rpw, &lds)) col = true;
        if (!t->resident_off && (col || resident_plan(t->R, t->C, t->ld, &grid, &rpw, &lds))) {
            const int rc = run_resident(t, o, cb, user, st, grid, rpw, lds, &resume, col);
            if (rc != LPX_RESIDENT_RETRY) return rc;
            if (o->resident > 0) return LPX_EDEVICE;        // required, and it could not run
        } else
        if (o->resident > 0) { set_error("lpx_primal_run: resident = 1 but the tableau does not fit the chip's LDS"); return LPX_EINVAL; }
    }
    SelParams p = base_params(t, o, MODE_PRIMAL);
    // without a per-pivot callback: one fused launch per pivot (LPX_FUSED_PIVOT=0: the two-launch in-place kernels)
    static const bool fused_env = [] { const char* e = std::getenv("LPX_FUSED_PIVOT"); return !(e && e[0] == '0'); }();
    if (fused_env && p.us && !cb && fused_buffers(t)) return run_fused(t, p, o, st, resume);
    return run_loop(t, p, o, (long long)o->max_iter + 2, cb, user, st, resume);
}

int lpx_dual_run(lpx_tableau* t, const lpx_run_opts* o, lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    if (!t) { set_error("lpx_dual_run: null tableau"); return LPX_EINVAL; }
    lpx_run_opts d; if (!o) { lpx_default_opts(&d, 1); o = &d; }
    if (t->R < 2) { set_error("lpx_dual_run: tableau needs at least one constraint row"); return LPX_EINVAL; }
    static const bool res_env = [] { const char* e = std::getenv("LPX_RESIDENT"); return !(e && e[0] == '0'); }();
    if (o->resident > 0 || (o->resident == 0 && res_env && !o->profile && (o->batch == 0 || o->batch >= 32))) {
        int grid = 0, slots = 0; size_t lds = 0; int isdual = 1, status = 0;
        if (!t->resident_off && resident_group_plan(&t, 1, &grid, &slots, &lds)) {
            DevState resume;
            const int rc = run_resident_group(&t, &isdual, 1, o, o, &status, st, grid, slots, lds, cb, user, &resume);
            if (rc == 0) return status;
            if (rc != LPX_RESIDENT_RETRY) return rc;
            t->resident_off = true;
            if (o->resident > 0) return LPX_EDEVICE;
            // hand over to the streaming kernels at the pivot the resident loop had reached
            SelParams p = base_params(t, o, MODE_DUAL);
            LoopCtx c; DevState init;
            make_ctx(t, p, c, init);
            init.iter = resume.iter; init.phase = resume.phase;
            init.fdf_count = resume.fdf_count; init.dual_iter = resume.dual_iter; init.primal_count = resume.primal_count;
            c.start_iter = resume.iter;
            return run_device_loop(c, init, o, (long long)o->fdf_guard + 2LL * o->max_iter + 8, cb, user, st);
        } else
        if (o->resident > 0) { set_error("lpx_dual_run: resident = 1 but the tableau does not fit the chip's LDS"); return LPX_EINVAL; }
    }
    if (!cb) {          // no per-pivot callback: one launch per step (lpx_group_fused, a group of one), update out of place
        int isdual = 1, status = 0;
        const int rc = multi_run_fused(&t, &isdual, 1, o, o, &status, st, nullptr, 0);
        if (rc != LPX_RESIDENT_RETRY) return rc ? rc : status;
    }
    SelParams p = base_params(t, o, MODE_DUAL);
    long long budget = (long long)o->fdf_guard + 2LL * o->max_iter + 8;
    return run_loop(t, p, o, budget, cb, user, st);
}

int lpx_forced_pivots_run(lpx_tableau* t, const int32_t* rows, const int32_t* cols, int count,
                          double thresh, int32_t* chosen, const lpx_run_opts* o, lpx_stats* st)
{
    if (!t || !rows || !cols || count < 0) { set_error("lpx_forced_pivots_run: bad argument"); return LPX_EINVAL; }
    lpx_run_opts d; if (!o) { lpx_default_opts(&d, 0); o = &d; }
    for (int k = 0; k < count; ++k)
        if (rows[k] < 0 || rows[k] >= t->R || cols[k] < 0 || cols[k] >= t->C) {
            set_error("lpx_forced_pivots_run: pivot position outside the tableau"); return LPX_EINVAL;
        }
    if (count > t->fcap) {
        hipFree(t->frows); hipFree(t->fcols); hipFree(t->fchosen);
        t->frows = t->fcols = t->fchosen = nullptr; t->fcap = 0;
        LPX_HIP_TRY(hipMalloc((void**)&t->frows, sizeof(int32_t) * count));
        LPX_HIP_TRY(hipMalloc((void**)&t->fcols, sizeof(int32_t) * count));
        LPX_HIP_TRY(hipMalloc((void**)&t->fchosen, sizeof(int32_t) * count));
        t->fcap = count;
        drop_graph(t);
    }
    if (count > 0) {
        LPX_HIP_TRY(hipMemcpyAsync(t->frows, rows, sizeof(int32_t) * count, hipMemcpyHostToDevice, t->stream));
        LPX_HIP_TRY(hipMemcpyAsync(t->fcols, cols, sizeof(int32_t) * count, hipMemcpyHostToDevice, t->stream));
        LPX_HIP_TRY(hipMemsetAsync(t->fchosen, 0xff, sizeof(int32_t) * count, t->stream));
    }
    SelParams p = base_params(t, o, MODE_FORCED);
    p.frows = t->frows; p.fcols = t->fcols; p.fcount = count; p.fthresh = thresh; p.fchosen = t->fchosen;
    int rc = run_loop(t, p, o, (long long)count + 2, nullptr, nullptr, st);
    if (chosen && count > 0)
        LPX_HIP_TRY(hipMemcpy(chosen, t->fchosen, sizeof(int32_t) * count, hipMemcpyDeviceToHost));
    return rc;
}

static int one_shot(double* T, int R, int C, int32_t* basis, const lpx_run_opts* o, int dual,
                    lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    if (!T || !basis) { set_error("null tableau or basis"); return LPX_EINVAL; }
    lpx_tableau* t = nullptr;
    int rc = lpx_tableau_create(R, C, &t);
    if (rc) return rc;
    double a = now_ms();
    rc = lpx_tableau_upload(t, T, basis);
    double h2d = now_ms() - a;
    if (rc) { lpx_tableau_destroy(t); return rc; }
    lpx_stats local; std::memset(&local, 0, sizeof(local));
    int status = dual ? lpx_dual_run(t, o, cb, user, &local) : lpx_primal_run(t, o, cb, user, &local);
    if (status < 0) { lpx_tableau_destroy(t); return status; }
    a = now_ms();
    rc = lpx_tableau_download(t, T, basis);
    local.d2h_ms = now_ms() - a;
    local.h2d_ms = h2d;
    lpx_tableau_destroy(t);
    if (st) *st = local;
    return rc ? rc : status;
}

int lpx_tableau_set_shape(lpx_tableau* t, int R, int C)
{
    if (!t || R < 1 || C < 2 || R > t->Rcap || C > t->Ccap) { set_error("lpx_tableau_set_shape: shape outside the handle's capacity"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));          // the pinned staging word may still be in flight
    t->R = R; t->C = C; t->suspended = t->suspended2 = t->fsuspended = false;               // a new tableau is coming: nothing to continue
    t->shape_h[0] = R; t->shape_h[1] = C;
    LPX_HIP_TRY(hipMemcpyAsync(t->shape, t->shape_h, sizeof(int32_t) * 2, hipMemcpyHostToDevice, t->stream));
    return 0;
}

int lpx_tableau_build_node(lpx_tableau* node, const lpx_tableau* root, int ncuts, const int32_t* var,
                           const double* coef, const double* zero, const double* rhs)
{
    if (!node || !root || ncuts < 0 || (ncuts > 0 && (!var || !coef || !zero || !rhs))) { set_error("lpx_tableau_build_node: bad argument"); return LPX_EINVAL; }
    if (root->R + ncuts > node->Rcap || root->C + ncuts > node->Ccap) { set_error("lpx_tableau_build_node: root shape + ncuts exceeds the node handle's capacity"); return LPX_EINVAL; }
    { int rc = lpx_tableau_set_shape(node, root->R + ncuts, root->C + ncuts); if (rc) return rc; }
    const int n = root->C - root->R;
    for (int k = 0; k < ncuts; ++k) if (var[k] < 0 || var[k] >= n) { set_error("lpx_tableau_build_node: branching variable out of range"); return LPX_EINVAL; }
    const int need = ncuts > 0 ? ncuts : 1;
    if (need > node->cutcap) {
        hipFree(node->cutbuf); if (node->cutbuf_h) hipHostFree(node->cutbuf_h);
        node->cutbuf = nullptr; node->cutbuf_h = nullptr; node->cutcap = 0;
        const int c = need + 64;
        LPX_HIP_TRY(hipMalloc((void**)&node->cutbuf, (size_t)c * 32));
        LPX_HIP_TRY(hipHostMalloc((void**)&node->cutbuf_h, (size_t)c * 32));
        node->cutcap = c;
    }
    const size_t cap = (size_t)node->cutcap;
    double* hc = reinterpret_cast<double*>(node->cutbuf_h);          // [coef | zero | rhs | var(int32, padded)]
    for (int k = 0; k < ncuts; ++k) { hc[k] = coef[k]; hc[cap + k] = zero[k]; hc[2 * cap + k] = rhs[k]; }
    int32_t* hv = reinterpret_cast<int32_t*>(hc + 3 * cap);
    for (int k = 0; k < ncuts; ++k) hv[k] = var[k];
    LPX_HIP_TRY(hipMemcpyAsync(node->cutbuf, node->cutbuf_h, cap * 32, hipMemcpyHostToDevice, node->stream));
    const double* dc = reinterpret_cast<const double*>(node->cutbuf);
    const double* T0 = root->snapT ? root->snapT : root->T;          // the pristine root tableau
    LPX_HIP_TRY(launch_build_node(T0, root->ld, root->R, root->C, node->T, node->ld, node->R, node->C,
                                  reinterpret_cast<const int32_t*>(dc + 3 * cap), dc, dc + cap, dc + 2 * cap,
                                  node->basis, node->stream));
    LPX_HIP_TRY(hipMemsetAsync(node->st, 0, sizeof(DevState), node->stream));
    return 0;   // stream-ordered: the run that follows on node->stream sees the finished tableau
}

// lpx_tableau_build_node for a group of nodes in one launch: node i gets the cuts [cut_off[i], cut_off[i + 1]) of the
// flattened arrays.  One H2D copy of the packed descriptors and cuts, one kernel, one wait.
int lpx_tableau_build_nodes(lpx_tableau** nodes, const lpx_tableau* root, int count, const int32_t* cut_off,
                            const int32_t* var, const double* coef, const double* zero, const double* rhs)
{
    if (!nodes || !root || count < 0 || !cut_off) { set_error("lpx_tableau_build_nodes: bad argument"); return LPX_EINVAL; }
    if (count == 0) return 0;
    const int total = cut_off[count];
    if (cut_off[0] != 0 || total < 0 || (total > 0 && (!var || !coef || !zero || !rhs))) { set_error("lpx_tableau_build_nodes: bad cut arrays"); return LPX_EINVAL; }
    const int n = root->C - root->R;
    int maxld = 16, maxR = 1;
    for (int i = 0; i < count; ++i) {
        lpx_tableau* t = nodes[i];
        const int nc = cut_off[i + 1] - cut_off[i];
        if (!t || nc < 0) { set_error("lpx_tableau_build_nodes: null node or negative cut count"); return LPX_EINVAL; }
        if (root->R + nc > t->Rcap || root->C + nc > t->Ccap) { set_error("lpx_tableau_build_nodes: root shape + ncuts exceeds a node handle's capacity"); return LPX_EINVAL; }
        maxld = std::max(maxld, t->ld); maxR = std::max(maxR, root->R + nc);
    }
    for (int k = 0; k < total; ++k) if (var[k] < 0 || var[k] >= n) { set_error("lpx_tableau_build_nodes: branching variable out of range"); return LPX_EINVAL; }
    struct Scratch { char* h = nullptr; char* d = nullptr; size_t cap = 0; };
    static thread_local Scratch sc;             // never freed: the HIP runtime may be gone when thread-locals are torn down
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t tn = (size_t)(total > 0 ? total : 1);
    const size_t o_desc = 0, o_coef = up16(sizeof(BuildDesc) * (size_t)count), o_zero = o_coef + up16(8 * tn), o_rhs = o_zero + up16(8 * tn),
                 o_var = o_rhs + up16(8 * tn), need = o_var + up16(4 * tn);
    if (need > sc.cap) {
        if (sc.h) hipHostFree(sc.h);
        hipFree(sc.d);
        sc.h = nullptr; sc.d = nullptr; sc.cap = 0;
        LPX_HIP_TRY(hipHostMalloc((void**)&sc.h, 2 * need));
        LPX_HIP_TRY(hipMalloc((void**)&sc.d, 2 * need));
        sc.cap = 2 * need;
    }
    hipStream_t s = nodes[0]->stream;
    // every node's own stream has to be idle before another stream writes its tableau (and before the scratch is reused)
    for (int i = 0; i < count; ++i) {
        bool seen = false; for (int j = 0; j < i; ++j) if (nodes[j]->stream == nodes[i]->stream) { seen = true; break; }
        if (!seen) LPX_HIP_TRY(hipStreamSynchronize(nodes[i]->stream));
    }
    BuildDesc* d = reinterpret_cast<BuildDesc*>(sc.h + o_desc);
    for (int i = 0; i < count; ++i) {
        lpx_tableau* t = nodes[i];
        const int nc = cut_off[i + 1] - cut_off[i];
        t->R = root->R + nc; t->C = root->C + nc; t->suspended = t->suspended2 = t->fsuspended = false;
        t->shape_h[0] = t->R; t->shape_h[1] = t->C;
        d[i].T = t->T; d[i].basis = t->basis; d[i].shape = t->shape; d[i].st = t->st; d[i].ld = t->ld; d[i].R = t->R; d[i].C = t->C; d[i].cut0 = cut_off[i];
    }
    if (total > 0) {
        std::memcpy(sc.h + o_coef, coef, 8 * (size_t)total); std::memcpy(sc.h + o_zero, zero, 8 * (size_t)total);
        std::memcpy(sc.h + o_rhs, rhs, 8 * (size_t)total); std::memcpy(sc.h + o_var, var, 4 * (size_t)total);
    }
    LPX_HIP_TRY(hipMemcpyAsync(sc.d, sc.h, need, hipMemcpyHostToDevice, s));
    const double* T0 = root->snapT ? root->snapT : root->T;          // the pristine root tableau
    LPX_HIP_TRY(launch_build_nodes(T0, root->ld, root->R, root->C, reinterpret_cast<const BuildDesc*>(sc.d + o_desc), count, maxld, maxR,
                                   reinterpret_cast<const int32_t*>(sc.d + o_var), reinterpret_cast<const double*>(sc.d + o_coef),
                                   reinterpret_cast<const double*>(sc.d + o_zero), reinterpret_cast<const double*>(sc.d + o_rhs), s));
    LPX_HIP_TRY(hipStreamSynchronize(s));       // the runs that follow use other streams
    return 0;
}

// ---- parent store: final tableaux of solved nodes parked in slab slots (warm-started B&B children) ----------
// Chunks of a destroyed store are kept for the next one (per process, up to LPX_STORE_CACHE_GB, default 64): a warm-started
// search parks thousands of parent tableaux and grows its store by 1 GB allocations, which cost a search that follows other
// GPU work on the same box up to half its time (bench.py's warm B&B leg right after the GPU test suite: 4.5-4.8 k nodes/s
// against 6.8-7.2 k; hipMalloc of memory another process has just released).
namespace {
std::mutex g_chunk_mu;
std::multimap<size_t, void*> g_chunk_cache;
size_t g_chunk_cached = 0;
size_t chunk_cache_max()
{
    // LPX_STORE_CACHE_GB (clamped to >= 0; default 64).  Whatever the cap, a chunk is only kept while a quarter of the device's memory
    // stays free without it (chunk_release): several ranks sharing one GPU, or a big tableau allocated next, must not find the
    // memory sitting idle in here -- and every large allocation of the library retries once after lpx::trim_device_caches().
    static const size_t v = [] { const char* e = std::getenv("LPX_STORE_CACHE_GB"); long g = e ? std::atol(e) : 64; if (g < 0) g = 0; if (g > 4096) g = 4096; return (size_t)g << 30; }();
    return v;
}
void chunk_cache_drop_all()
{
    std::lock_guard<std::mutex> lk(g_chunk_mu);
    for (auto& kv : g_chunk_cache) hipFree(kv.second);
    g_chunk_cache.clear(); g_chunk_cached = 0;
}
hipError_t chunk_alloc(void** p, size_t bytes)
{
    {
        std::lock_guard<std::mutex> lk(g_chunk_mu);
        auto it = g_chunk_cache.find(bytes);
        if (it != g_chunk_cache.end()) { *p = it->second; g_chunk_cache.erase(it); g_chunk_cached -= bytes; return hipSuccess; }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    // out of memory with chunks of other sizes in the cache: give them back and try once more
    (void)hipGetLastError();
    chunk_cache_drop_all();
    return hipMalloc(p, bytes);
}
void chunk_release(void* p, size_t bytes)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_chunk_mu);
        size_t free_b = 0, total_b = 0;
        const bool roomy = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= total_b / 4;
        if (roomy && g_chunk_cached + bytes <= chunk_cache_max()) { g_chunk_cache.emplace(bytes, p); g_chunk_cached += bytes; return; }
    }
    hipFree(p);
}
}  // namespace

extern "C++" {
namespace lpx {
void trim_device_caches() { chunk_cache_drop_all(); }
hipError_t malloc_retry(void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    trim_device_caches();
    return hipMalloc(p, bytes);
}
}  // namespace lpx
}  // extern "C++"

struct lpx_store {
    int Rcap = 0, Ccap = 0, ld = 0, per_chunk = 128;
    size_t slot_doubles = 0;                 // Rcap * ld
    std::vector<double*> chunks_T; std::vector<int32_t*> chunks_b;
    std::vector<int> R, C;                   // live shape per slot
    std::vector<int> free_slots;
};

int lpx_store_create(int Rcap, int Ccap, lpx_store** out)
{
    if (!out || Rcap < 2 || Ccap < 2) { set_error("lpx_store_create: bad shape"); return LPX_EINVAL; }
    int rc = ensure_device(); if (rc) return rc;
    lpx_store* s = new lpx_store();
    s->Rcap = Rcap; s->Ccap = Ccap; s->ld = (Ccap + 15) / 16 * 16;
    s->slot_doubles = (size_t)Rcap * s->ld;
    *out = s;
    return 0;
}

void lpx_store_destroy(lpx_store* s)
{
    if (!s) return;
    // a chunk that enters the cache may be handed to another store at once: nothing (a parking copy, a child assembly reading a
    // parked parent -- they run on the handles' streams) may still be using it.  hipFree used to give this wait for free.
    if (!s->chunks_T.empty()) (void)hipDeviceSynchronize();
    for (double* p : s->chunks_T) chunk_release(p, sizeof(double) * s->slot_doubles * s->per_chunk);
    for (int32_t* p : s->chunks_b) chunk_release(p, sizeof(int32_t) * (size_t)s->Rcap * s->per_chunk);
    delete s;
}

static double* store_T(lpx_store* s, int slot) { return s->chunks_T[slot / s->per_chunk] + (size_t)(slot % s->per_chunk) * s->slot_doubles; }
static int32_t* store_b(lpx_store* s, int slot) { return s->chunks_b[slot / s->per_chunk] + (size_t)(slot % s->per_chunk) * s->Rcap; }

int lpx_store_save(lpx_store* s, lpx_tableau* t, int* slot_out)
{
    if (!s || !t || !slot_out) { set_error("lpx_store_save: null argument"); return LPX_EINVAL; }
    if (t->ld != s->ld || t->R > s->Rcap) { set_error("lpx_store_save: tableau does not match the store's capacity class"); return LPX_EINVAL; }
    if (s->free_slots.empty()) {
        double* Tc = nullptr; int32_t* bc = nullptr;
        LPX_HIP_TRY(chunk_alloc((void**)&Tc, sizeof(double) * s->slot_doubles * s->per_chunk));
        hipError_t e = chunk_alloc((void**)&bc, sizeof(int32_t) * (size_t)s->Rcap * s->per_chunk);
        if (e != hipSuccess) { hipFree(Tc); set_error("lpx_store_save: out of device memory"); return LPX_ENOMEM; }
        const int base = (int)s->chunks_T.size() * s->per_chunk;
        s->chunks_T.push_back(Tc); s->chunks_b.push_back(bc);
        s->R.resize(base + s->per_chunk, 0); s->C.resize(base + s->per_chunk, 0);
        for (int k = s->per_chunk - 1; k >= 0; --k) s->free_slots.push_back(base + k);
    }
    const int slot = s->free_slots.back(); s->free_slots.pop_back();
    LPX_HIP_TRY(hipMemcpyAsync(store_T(s, slot), t->T, sizeof(double) * (size_t)t->R * t->ld, hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipMemcpyAsync(store_b(s, slot), t->basis, sizeof(int32_t) * (t->R - 1), hipMemcpyDeviceToDevice, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));       // the handle may be reused by the caller right away
    s->R[slot] = t->R; s->C[slot] = t->C;
    *slot_out = slot;
    return 0;
}

int lpx_store_release(lpx_store* s, int slot)
{
    if (!s || slot < 0 || slot >= (int)s->R.size()) return LPX_EINVAL;
    s->free_slots.push_back(slot);
    return 0;
}

int lpx_tableau_build_child_from_store(lpx_tableau* child, lpx_store* s, int slot, int var, int row_of_var, int is_ge, double bound)
{
    if (!child || !s || slot < 0 || slot >= (int)s->R.size()) { set_error("lpx_tableau_build_child_from_store: bad argument"); return LPX_EINVAL; }
    const int Rp = s->R[slot], Cp = s->C[slot];
    if (var < 0 || var >= Cp - 1 || row_of_var < 0 || row_of_var >= Rp - 1) { set_error("lpx_tableau_build_child_from_store: variable / row out of range"); return LPX_EINVAL; }
    if (Rp + 1 > child->Rcap || Cp + 1 > child->Ccap) { set_error("lpx_tableau_build_child_from_store: child handle too small"); return LPX_EINVAL; }
    { int rc = lpx_tableau_set_shape(child, Rp + 1, Cp + 1); if (rc) return rc; }
    LPX_HIP_TRY(launch_build_child(store_T(s, slot), s->ld, Rp, Cp, store_b(s, slot), child->T, child->ld,
                                   var, row_of_var, is_ge ? 1 : 0, bound, child->basis, child->stream));
    LPX_HIP_TRY(hipMemsetAsync(child->st, 0, sizeof(DevState), child->stream));
    return 0;
}

// lpx_tableau_build_child_from_store for a group of children in one launch (child i from stores[i] / slots[i]).
int lpx_tableau_build_children_from_store(lpx_tableau** children, lpx_store** stores, const int* slots, int count,
                                          const int32_t* var, const int32_t* row_of_var, const int32_t* is_ge, const double* bound)
{
    if (!children || !stores || !slots || count < 0 || (count > 0 && (!var || !row_of_var || !is_ge || !bound))) { set_error("lpx_tableau_build_children_from_store: bad argument"); return LPX_EINVAL; }
    if (count == 0) return 0;
    struct Scratch { char* h = nullptr; char* d = nullptr; size_t cap = 0; };
    static thread_local Scratch sc;             // never freed (see lpx_tableau_build_nodes)
    const size_t need = sizeof(ChildDesc) * (size_t)count;
    if (need > sc.cap) {
        if (sc.h) hipHostFree(sc.h);
        hipFree(sc.d);
        sc.h = nullptr; sc.d = nullptr; sc.cap = 0;
        LPX_HIP_TRY(hipHostMalloc((void**)&sc.h, 2 * need));
        LPX_HIP_TRY(hipMalloc((void**)&sc.d, 2 * need));
        sc.cap = 2 * need;
    }
    int maxld = 16, maxR = 1;
    for (int i = 0; i < count; ++i) {
        lpx_tableau* ch = children[i]; lpx_store* s = stores[i]; const int slot = slots[i];
        if (!ch || !s || slot < 0 || slot >= (int)s->R.size()) { set_error("lpx_tableau_build_children_from_store: bad child / store / slot"); return LPX_EINVAL; }
        const int Rp = s->R[slot], Cp = s->C[slot];
        if (var[i] < 0 || var[i] >= Cp - 1 || row_of_var[i] < 0 || row_of_var[i] >= Rp - 1) { set_error("lpx_tableau_build_children_from_store: variable / row out of range"); return LPX_EINVAL; }
        if (Rp + 1 > ch->Rcap || Cp + 1 > ch->Ccap) { set_error("lpx_tableau_build_children_from_store: child handle too small"); return LPX_EINVAL; }
        maxld = std::max(maxld, ch->ld); maxR = std::max(maxR, Rp + 1);
    }
    for (int i = 0; i < count; ++i) {           // every child's own stream has to be idle before another stream writes its tableau
        bool seen = false; for (int j = 0; j < i; ++j) if (children[j]->stream == children[i]->stream) { seen = true; break; }
        if (!seen) LPX_HIP_TRY(hipStreamSynchronize(children[i]->stream));
    }
    ChildDesc* d = reinterpret_cast<ChildDesc*>(sc.h);
    for (int i = 0; i < count; ++i) {
        lpx_tableau* ch = children[i]; lpx_store* s = stores[i]; const int slot = slots[i];
        const int Rp = s->R[slot], Cp = s->C[slot];
        ch->R = Rp + 1; ch->C = Cp + 1; ch->suspended = ch->suspended2 = ch->fsuspended = false; ch->shape_h[0] = ch->R; ch->shape_h[1] = ch->C;
        d[i].Tp = store_T(s, slot); d[i].basis_p = store_b(s, slot); d[i].T = ch->T; d[i].basis = ch->basis; d[i].shape = ch->shape; d[i].st = ch->st;
        d[i].ldp = s->ld; d[i].Rp = Rp; d[i].Cp = Cp; d[i].ld = ch->ld; d[i].var = var[i]; d[i].ik = row_of_var[i]; d[i].is_ge = is_ge[i] ? 1 : 0; d[i].pad = 0;
        d[i].bound = bound[i];
    }
    hipStream_t st = children[0]->stream;
    LPX_HIP_TRY(hipMemcpyAsync(sc.d, sc.h, need, hipMemcpyHostToDevice, st));
    LPX_HIP_TRY(launch_build_children(reinterpret_cast<const ChildDesc*>(sc.d), count, maxld, maxR, st));
    LPX_HIP_TRY(hipStreamSynchronize(st));      // the runs that follow use other streams
    return 0;
}

int lpx_tableau_build_child(lpx_tableau* child, lpx_tableau* parent, int var, int row_of_var, int is_ge, double bound)
{
    if (!child || !parent || child == parent) { set_error("lpx_tableau_build_child: bad argument"); return LPX_EINVAL; }
    if (var < 0 || var >= parent->C - 1 || row_of_var < 0 || row_of_var >= parent->R - 1) { set_error("lpx_tableau_build_child: variable / row out of range"); return LPX_EINVAL; }
    if (parent->R + 1 > child->Rcap || parent->C + 1 > child->Ccap) { set_error("lpx_tableau_build_child: child handle too small"); return LPX_EINVAL; }
    LPX_HIP_TRY(hipStreamSynchronize(parent->stream));              // the parent's final tableau must be complete
    { int rc = lpx_tableau_set_shape(child, parent->R + 1, parent->C + 1); if (rc) return rc; }
    LPX_HIP_TRY(launch_build_child(parent->T, parent->ld, parent->R, parent->C, parent->basis, child->T, child->ld,
                                   var, row_of_var, is_ge ? 1 : 0, bound, child->basis, child->stream));
    LPX_HIP_TRY(hipMemsetAsync(child->st, 0, sizeof(DevState), child->stream));
    return 0;
}

int lpx_tableau_basis(lpx_tableau* t, int32_t* basis)
{
    if (!t || !basis) return LPX_EINVAL;
    if (t->R > 1) LPX_HIP_TRY(hipMemcpyAsync(basis, t->basis, sizeof(int32_t) * (t->R - 1), hipMemcpyDeviceToHost, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    return 0;
}

int lpx_tableau_solution2(lpx_tableau* t, int nvars, double* x, double* z, int32_t* basis_out);
int lpx_tableau_solution(lpx_tableau* t, int nvars, double* x, double* z) { return lpx_tableau_solution2(t, nvars, x, z, nullptr); }

int lpx_tableau_solution2(lpx_tableau* t, int nvars, double* x, double* z, int32_t* basis_out)
{
    if (!t || nvars < 0) { set_error("lpx_tableau_solution: bad argument"); return LPX_EINVAL; }
    const int m = t->R - 1;
    std::vector<double> rhs(t->R);
    std::vector<int32_t> basis(m > 0 ? m : 1);
    LPX_HIP_TRY(hipMemcpy2DAsync(rhs.data(), sizeof(double), t->T + (t->C - 1), sizeof(double) * t->ld,
                                 sizeof(double), t->R, hipMemcpyDeviceToHost, t->stream));
    if (m > 0) LPX_HIP_TRY(hipMemcpyAsync(basis.data(), t->basis, sizeof(int32_t) * m, hipMemcpyDeviceToHost, t->stream));
    LPX_HIP_TRY(hipStreamSynchronize(t->stream));
    if (x) {
        for (int j = 0; j < nvars; ++j) x[j] = 0.0;
        for (int i = 0; i < m; ++i) if (basis[i] >= 0 && basis[i] < nvars) x[basis[i]] = rhs[i];   // FinalizeReport :135-136
    }
    if (z) *z = rhs[m];                                                                           // :138
    if (basis_out && m > 0) std::memcpy(basis_out, basis.data(), sizeof(int32_t) * m);
    return 0;
}

// lpx_tableau_solution2 for a batch: x is count x nvars, z has count entries, basis_out (optional) count x basis_stride.
int lpx_multi_solution(lpx_tableau** ts, int count, int nvars, double* x, double* z, int32_t* basis_out, int basis_stride)
{
    if (!ts || count < 0 || nvars < 0) { set_error("lpx_multi_solution: bad argument"); return LPX_EINVAL; }
    if (count == 0) return 0;
    struct Scratch { char* h = nullptr; size_t cap = 0; ~Scratch() { if (h) hipHostFree(h); } };
    static thread_local Scratch sc;
    size_t rows = 0;
    for (int i = 0; i < count; ++i) {
        if (!ts[i] || ts[i]->R < 1) { set_error("lpx_multi_solution: null or empty tableau"); return LPX_EINVAL; }
        if (basis_out && basis_stride < ts[i]->R - 1) { set_error("lpx_multi_solution: basis_stride too small"); return LPX_EINVAL; }
        rows += (size_t)ts[i]->R;
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_desc = 0, o_rhs = up16(sizeof(GatherDesc) * (size_t)count), o_bas = o_rhs + up16(sizeof(double) * rows);
    const size_t need = o_bas + up16(sizeof(int32_t) * rows);
    if (need > sc.cap) {
        if (sc.h) hipHostFree(sc.h);
        sc.h = nullptr; sc.cap = 0;
        LPX_HIP_TRY(hipHostMalloc((void**)&sc.h, 2 * need));
        sc.cap = 2 * need;
    }
    GatherDesc* d = reinterpret_cast<GatherDesc*>(sc.h + o_desc);
    double* rhs = reinterpret_cast<double*>(sc.h + o_rhs);
    int32_t* bas = reinterpret_cast<int32_t*>(sc.h + o_bas);
    size_t off = 0;
    for (int i = 0; i < count; ++i) {
        d[i].T = ts[i]->T; d[i].basis = ts[i]->basis; d[i].ld = ts[i]->ld; d[i].R = ts[i]->R; d[i].C = ts[i]->C; d[i].off = (int)off;
        off += (size_t)ts[i]->R;
    }
    // every handle's own stream has to be idle before another stream reads its tableau
    for (int i = 0; i < count; ++i) {
        bool seen = false; for (int j = 0; j < i; ++j) if (ts[j]->stream == ts[i]->stream) { seen = true; break; }
        if (!seen) LPX_HIP_TRY(hipStreamSynchronize(ts[i]->stream));
    }
    hipStream_t s = ts[0]->stream;
    LPX_HIP_TRY(launch_gather_solution(d, count, rhs, bas, s));
    LPX_HIP_TRY(hipStreamSynchronize(s));
    for (int i = 0; i < count; ++i) {
        const int m = ts[i]->R - 1;
        const double* r = rhs + d[i].off; const int32_t* b = bas + d[i].off;
        if (x) {
            double* xi = x + (size_t)i * nvars;
            for (int j = 0; j < nvars; ++j) xi[j] = 0.0;
            for (int k = 0; k < m; ++k) if (b[k] >= 0 && b[k] < nvars) xi[b[k]] = r[k];       // FinalizeReport :135-136
        }
        if (z) z[i] = r[m];                                                                 // :138
        if (basis_out && m > 0) std::memcpy(basis_out + (size_t)i * basis_stride, b, sizeof(int32_t) * m);
    }
    return 0;
}

// lpx_store_save for a batch: all copies are enqueued first, each stream is waited for once.
int lpx_store_save_multi(lpx_store** ss, lpx_tableau** ts, int count, int* slots)
{
    if (!ss || !ts || !slots || count < 0) { set_error("lpx_store_save_multi: bad argument"); return LPX_EINVAL; }
    if (count == 0) return 0;
    for (int i = 0; i < count; ++i) {
        lpx_store* s = ss[i]; lpx_tableau* t = ts[i];
        if (!s || !t) { set_error("lpx_store_save_multi: null argument"); return LPX_EINVAL; }
        if (t->ld != s->ld || t->R > s->Rcap) { set_error("lpx_store_save_multi: tableau does not match the store's capacity class"); return LPX_EINVAL; }
    }
    struct Scratch { char* h = nullptr; char* d = nullptr; size_t cap = 0; };
    static thread_local Scratch sc;             // never freed (see lpx_tableau_build_nodes)
    const size_t need = sizeof(ParkDesc) * (size_t)count;
    if (need > sc.cap) {
        if (sc.h) hipHostFree(sc.h);
        hipFree(sc.d);
        sc.h = nullptr; sc.d = nullptr; sc.cap = 0;
        LPX_HIP_TRY(hipHostMalloc((void**)&sc.h, 2 * need));
        LPX_HIP_TRY(hipMalloc((void**)&sc.d, 2 * need));
        sc.cap = 2 * need;
    }
    for (int i = 0; i < count; ++i) {           // the finished runs used the group's stream; the nodes' own streams are idle, make sure
        bool seen = false; for (int j = 0; j < i; ++j) if (ts[j]->stream == ts[i]->stream) { seen = true; break; }
        if (!seen) LPX_HIP_TRY(hipStreamSynchronize(ts[i]->stream));
    }
    ParkDesc* d = reinterpret_cast<ParkDesc*>(sc.h);
    size_t maxd = 0;
    for (int i = 0; i < count; ++i) {
        lpx_store* s = ss[i]; lpx_tableau* t = ts[i];
        if (s->free_slots.empty()) {
            double* Tc = nullptr; int32_t* bc = nullptr;
            LPX_HIP_TRY(chunk_alloc((void**)&Tc, sizeof(double) * s->slot_doubles * s->per_chunk));
            hipError_t e = chunk_alloc((void**)&bc, sizeof(int32_t) * (size_t)s->Rcap * s->per_chunk);
            if (e != hipSuccess) { hipFree(Tc); set_error("lpx_store_save_multi: out of device memory"); return LPX_ENOMEM; }
            const int base = (int)s->chunks_T.size() * s->per_chunk;
            s->chunks_T.push_back(Tc); s->chunks_b.push_back(bc);
            s->R.resize(base + s->per_chunk, 0); s->C.resize(base + s->per_chunk, 0);
            for (int k = s->per_chunk - 1; k >= 0; --k) s->free_slots.push_back(base + k);
        }
        const int slot = s->free_slots.back(); s->free_slots.pop_back();
        s->R[slot] = t->R; s->C[slot] = t->C;
        slots[i] = slot;
        d[i].srcT = t->T; d[i].dstT = store_T(s, slot); d[i].srcB = t->basis; d[i].dstB = store_b(s, slot);
        d[i].doubles = (size_t)t->R * t->ld; d[i].m = t->R - 1; d[i].pad = 0;
        maxd = std::max(maxd, d[i].doubles);
    }
    hipStream_t st = ts[0]->stream;
    LPX_HIP_TRY(hipMemcpyAsync(sc.d, sc.h, need, hipMemcpyHostToDevice, st));
    const int bpn = (int)std::min<size_t>(256, std::max<size_t>(1, maxd / 2 / 256 / 4));        // ~4 double2 per lane at least
    LPX_HIP_TRY(launch_park_many(reinterpret_cast<const ParkDesc*>(sc.d), count, bpn, st));
    LPX_HIP_TRY(hipStreamSynchronize(st));      // the handles may be reused, the slots read, right away
    return 0;
}

int lpx_multi_run(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts,
                  const lpx_run_opts* dopts, int* statuses, lpx_stats* stats)
{
    if (!ts || !dual || count < 0 || !statuses) { set_error("lpx_multi_run: bad argument"); return LPX_EINVAL; }
    lpx_run_opts pd, dd;
    if (!popts) { lpx_default_opts(&pd, 0); popts = &pd; }
    if (!dopts) { lpx_default_opts(&dd, 1); dopts = &dd; }
    // Nodes small enough to live on chip a few at a time (lpx_resident_group.hip): four 8 MB nodes of config 4 run side by
    // side, 25 % faster than 32 of them streaming through HBM.  LPX_RESIDENT_GROUP=0 keeps the batched streaming run.
    static const bool resgroup_env = [] { const char* e = std::getenv("LPX_RESIDENT_GROUP"); return !(e && e[0] == '0'); }();
    if (resgroup_env && count >= 1 && !popts->profile && !dopts->profile && popts->resident >= 0 && dopts->resident >= 0) {
        bool ok = true;
        for (int i = 0; i < count; ++i) if (!ts[i] || ts[i]->R < 2 || ts[i]->resident_off || (!dual[i] && !ts[i]->us)) ok = false;
        int grid = 0, slots = 0; size_t lds = 0;
        if (ok && resident_group_plan(ts, count, &grid, &slots, &lds)) {
            std::vector<DevState> resume(count);
            const int rc = run_resident_group(ts, dual, count, popts, dopts, statuses, stats, grid, slots, lds, nullptr, nullptr, resume.data());
            if (rc != LPX_RESIDENT_RETRY) return rc;
            // some workgroup could not take part (GPU shared with another process?): finish the unfinished nodes on the
            // batched streaming kernels, from the pivot each of them had reached
            std::vector<lpx_tableau*> sub; std::vector<int> sdual, sidx; std::vector<DevState> sinit;
            for (int i = 0; i < count; ++i) {
                ts[i]->resident_off = true;
                if (statuses[i] == LPX_RUNNING) { sub.push_back(ts[i]); sdual.push_back(dual[i]); sidx.push_back(i); sinit.push_back(resume[i]); }
                else if (stats) { std::memset(&stats[i], 0, sizeof(lpx_stats)); stats[i].pivots = resume[i].iter; stats[i].fdf_pivots = resume[i].fdf_count; stats[i].cleanup_pivots = dual[i] ? resume[i].primal_count : 0; *ts[i]->hst = resume[i]; }
            }
            if (!sub.empty()) {
                std::vector<int> sst(sub.size()); std::vector<lpx_stats> sss(sub.size());
                int rc2;
                if (sub.size() >= 2) rc2 = multi_run_batched(sub.data(), sdual.data(), (int)sub.size(), popts, dopts, sst.data(), sss.data(), sinit.data());
                else {
                    // a single straggler: its own streaming loop, continuing at its pivot count
                    lpx_tableau* t = sub[0];
                    const lpx_run_opts* o = sdual[0] ? dopts : popts;
                    SelParams p = base_params(t, o, sdual[0] ? MODE_DUAL : MODE_PRIMAL);
                    LoopCtx c; DevState init;
                    make_ctx(t, p, c, init);
                    init.iter = sinit[0].iter; init.phase = sdual[0] ? sinit[0].phase : 2;
                    init.fdf_count = sinit[0].fdf_count; init.dual_iter = sinit[0].dual_iter; init.primal_count = sinit[0].primal_count;
                    const long long budget = sdual[0] ? (long long)o->fdf_guard + 2LL * o->max_iter + 8 : (long long)o->max_iter + 2;
                    rc2 = run_device_loop(c, init, o, budget, nullptr, nullptr, &sss[0]);
                    if (rc2 >= 0) { sst[0] = rc2; rc2 = 0; }
                }
                if (rc2) return rc2;
                for (size_t k = 0; k < sub.size(); ++k) { statuses[sidx[k]] = sst[k]; if (stats) stats[sidx[k]] = sss[k]; }
            }
            return 0;
        }
    }
    static const bool batched_env = [] { const char* e = std::getenv("LPX_BATCHED"); return !(e && e[0] == '0'); }();
    if (batched_env && count >= 1 && (popts->profile || dopts->profile)) {
        // profile mode: the fused group launch bracketed by HIP events (stats[0].update_ms_sum / update_launches are the group's)
        bool ok = true;
        for (int i = 0; i < count; ++i) if (!ts[i] || ts[i]->R < 2) ok = false;
        if (ok) { const int rc = multi_run_fused(ts, dual, count, popts, dopts, statuses, stats, nullptr, 0); if (rc != LPX_RESIDENT_RETRY) return rc; }
    }
    if (batched_env && count >= 2 && !popts->profile && !dopts->profile) {
        bool ok = true;
        for (int i = 0; i < count; ++i) if (!ts[i] || ts[i]->R < 2 || (!dual[i] && !ts[i]->us)) ok = false;
        if (ok) return multi_run_batched(ts, dual, count, popts, dopts, statuses, stats);
    }
    std::vector<LoopRun> runs(count);
    std::vector<char> active(count, 0);
    for (int i = 0; i < count; ++i) {
        lpx_tableau* t = ts[i];
        if (!t || t->R < 2) { set_error("lpx_multi_run: null or empty tableau"); return LPX_EINVAL; }
        const lpx_run_opts* o = dual[i] ? dopts : popts;
        SelParams p = base_params(t, o, dual[i] ? MODE_DUAL : MODE_PRIMAL);
        LoopCtx c; DevState init;
        make_ctx(t, p, c, init);
        long long budget = dual[i] ? (long long)o->fdf_guard + 2LL * o->max_iter + 8 : (long long)o->max_iter + 2;
        int rc = runs[i].begin(c, init, o, budget, nullptr, nullptr);
        if (rc) return rc;
        active[i] = 1;
    }
    int remaining = count;
    while (remaining > 0) {
        for (int i = 0; i < count; ++i) if (active[i]) { int rc = runs[i].submit(); if (rc) return rc; }
        for (int i = 0; i < count; ++i) if (active[i]) {
            int rc = runs[i].complete(); if (rc) return rc;
            if (runs[i].done()) {
                statuses[i] = runs[i].finish(stats ? &stats[i] : nullptr);
                active[i] = 0; --remaining;
            }
        }
    }
    return 0;
}

// lpx_multi_run for a ROLLING batch (warm-started B&B children need a few dozen pivots each, a few of them hundreds): the run
// stops as soon as at most `min_active` nodes are still running and reports them as LPX_RUNNING; the caller hands them in again
// together with fresh nodes and they continue where they stopped (pivot count, phase, counters: the handle remembers).  Without
// it a batch of 64 runs as long as its slowest node while the device idles at the per-step latency floor.  Streaming batched
// kernels only (callers that want the resident group kernel use lpx_multi_run); min_active = 0 runs everything to the end.
int lpx_multi_run_some(lpx_tableau** ts, const int* dual, int count, const lpx_run_opts* popts, const lpx_run_opts* dopts,
                       int* statuses, lpx_stats* stats, int min_active)
{
    if (!ts || !dual || count < 0 || !statuses) { set_error("lpx_multi_run_some: bad argument"); return LPX_EINVAL; }
    lpx_run_opts pd, dd;
    if (!popts) { lpx_default_opts(&pd, 0); popts = &pd; }
    if (!dopts) { lpx_default_opts(&dd, 1); dopts = &dd; }
    bool ok = count >= 1 && !popts->profile && !dopts->profile, resumed = false;
    for (int i = 0; i < count; ++i) {
        if (!ts[i] || ts[i]->R < 2) { set_error("lpx_multi_run_some: null or empty tableau"); return LPX_EINVAL; }
        if (!dual[i] && !ts[i]->us) ok = false;
        if (ts[i]->suspended) resumed = true;
    }
    if (!ok) {
        if (resumed) { set_error("lpx_multi_run_some: a suspended run cannot continue on this path"); return LPX_EINVAL; }
        return lpx_multi_run(ts, dual, count, popts, dopts, statuses, stats);
    }
    std::vector<DevState> inits(count);
    for (int i = 0; i < count; ++i) {
        DevState z; std::memset(&z, 0, sizeof(z));
        z.status = LPX_RUNNING; z.phase = dual[i] ? 0 : 2;
        inits[i] = ts[i]->suspended ? *ts[i]->hst : z;
        ts[i]->suspended = false;
    }
    return multi_run_batched(ts, dual, count, popts, dopts, statuses, stats, inits.data(), min_active < count ? min_active : 0);
}

int lpx_primal_tableau(double* T, int R, int C, int32_t* basis, double eps, int max_iter,
                       lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    lpx_run_opts o; lpx_default_opts(&o, 0);
    o.eps = eps; o.ratio_tol = eps; o.max_iter = max_iter;
    return one_shot(T, R, C, basis, &o, 0, cb, user, st);
}

int lpx_dual_tableau(double* T, int R, int C, int32_t* basis, double eps, double ratio_tol,
                     int fdf_guard, int max_iter, int cleanup,
                     lpx_pivot_cb cb, void* user, lpx_stats* st)
{
    lpx_run_opts o; lpx_default_opts(&o, 1);
    o.eps = eps; o.ratio_tol = ratio_tol; o.fdf_guard = fdf_guard; o.max_iter = max_iter; o.cleanup = cleanup;
    return one_shot(T, R, C, basis, &o, 1, cb, user, st);
}

}  // extern "C"
