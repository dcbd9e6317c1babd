// lpx_internal.h -- shared between the HIP kernels and the host-side C ABI of liblpx.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <functional>
#include "../../include/lpx.h"

namespace lpx {

// Device-resident loop state: one instance per tableau handle, written only by the select kernel.
struct DevState {
    int status;          // LPX_RUNNING until a terminal status is reached
    int iter;            // pivots completed (index of the next trace slot)
    int r, q;            // pivot chosen by the last select launch (r < 0: nothing to update)
    int phase;           // 0 = ForceDualFeasibility, 1 = dual loop, 2 = primal loop
    int fdf_count;       // pivots done in phase 0
    int dual_iter;       // pivots done in phase 1
    int primal_count;    // pivots done in phase 2
    int forced_k;        // next entry of the forced-pivot list
    int qn;              // lookahead: entering column of the NEXT pivot (-1 = none), see lpx_select_la
    int c0n;             // multi-workgroup select: start column of the next forced search
    int qn_valid;        // multi-workgroup select: 1 = `qn` above overrides the partial reduction
    int pad[4];          // pad[0]: revised path's Nidx order counter; pad[1]: resident loop abort flag
};

enum { MODE_PRIMAL = 0, MODE_DUAL = 1, MODE_FORCED = 2 };

struct SelParams {
    double* T; int ld; int R; int C;    // R, C: CAPACITY of the handle (grid sizing)
    const int32_t* shape;               // device record {R, C} of the tableau currently in the handle
    double* prow;        // [ld]  normalised pivot row  T[r,:]/T[r,q]
    double* pcol;        // [R]   pivot column snapshot T[:,q]        (dual path)
    double* col0;        // [R]   lookahead column buffers, ping-pong   (primal / forced path)
    double* col1;
    double* rhsbuf;      // [R]   contiguous copy of the RHS column
    int32_t* basis;      // [R-1]
    int32_t* trace;      // [2*trace_cap]
    int trace_cap;
    DevState* st;        // written by select (block 0), read by update and the host
    DevState* us;        // multi-workgroup select: written by update (block 0), read by select
    double* part_v; int32_t* part_i; int nblk;   // per-workgroup partial argmins of the lookahead scan
    int qsel;                                    // 1: select's last workgroup reduces them (streaming update forms), 0: every update wave does
    double eps;          // Eps
    double tol_fdf;      // ratio hysteresis in phase 0
    double tol_dual;     // ratio hysteresis in phase 1
    double tol_primal;   // ratio hysteresis in phase 2
    int max_iter, fdf_guard, cleanup, mode;
    const int32_t* frows; const int32_t* fcols; int fcount; double fthresh; int32_t* fchosen;
    double* ws;          // [max(R,C)] global scratch for ratios when they do not fit LDS
    int rcap;            // doubles of dynamic LDS available for ratios (0 = use ws)
};

// launchers (lpx_kernels.hip)
hipError_t launch_select(const SelParams& p, hipStream_t s);       // gather-based (dual path)
hipError_t launch_select_la(const SelParams& p, hipStream_t s);    // lookahead (primal / forced)
hipError_t launch_la_init(const SelParams& p, hipStream_t s);
// fac0/fac1: factor columns, chosen by pivot parity; nxt by-products go to the other one.
// e0/e1 non-null: bracket the dispatch with HIP events bound to the kernel (hipExtLaunchKernelGGL).
hipError_t launch_update(double* T, int ld, int R, int C, const int32_t* shape, const double* prow, double* fac0, double* fac1,
                         double* rhsbuf, const DevState* st, hipStream_t s,
                         hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// multi-workgroup protocol: select_mb (nblk workgroups) + update_mb (reduces the partials, commits `us`)
hipError_t launch_select_mb(const SelParams& p, hipStream_t s);
hipError_t launch_update_mb(const SelParams& p, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
int select_mb_blocks(int C);
// Fused pivot (primal loop without a per-pivot callback, lpx_pivot_fused / _c in lpx_kernels.hip): ONE launch applies pivot k
// out of place (buffer b -> buffer 1 - b) and, in its first `nblk` workgroups, selects pivot k + 1 from the tableau it reads.
// P.T / P.prow / P.rhsbuf are the buffers of index 0, the members below those of index 1; P.col0 / P.col1 the factor columns.
struct FusedParams {
    SelParams P;
    double* T1; double* prow1; double* rhs1;
    DevState* rec;       // two state records: a launch reads rec[par] and writes rec[1 - par]; pad[2] = launch count, pad[3] = tableau buffer
    int par;             // set per launch by launch_pivot_fused
};
int fused_policy(int ld, int R);             // 0 = default cache policy (lpx_pivot_fused_c), 2 = streaming mix, 1 = all nt
hipError_t launch_fused_init(const FusedParams& f, hipStream_t s);
hipError_t launch_pivot_fused(const FusedParams& f, int par, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// fused group step (lpx_group_fused / _c): one launch per step for a whole group of node LPs, see lpx_kernels.hip
hipError_t launch_group_fused_init(const FusedParams* arr, const int* fresh, int nfresh, const DevState* init, hipStream_t s);
hipError_t launch_group_fused_gather(const FusedParams* arr, int count, DevState* out, int* cur, hipStream_t s);
int group_fused_blocks(int ld, int R);
// comp: the group's device-side compaction record (group_fused_comp_ints(cap) ints; the host writes count and list of parity 0 before
// the first launch of a window, the kernel keeps them current from launch to launch)
int group_fused_comp_ints(int cap);
int group_fused_comp_hdr();     // ints in front of the list of a parity region: {count, padding}
hipError_t launch_group_fused(const FusedParams* arr, const int* live, int nlive, int per_node, int lpar, size_t live_bytes, hipStream_t s,
                              int* comp, int cap, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
hipError_t launch_states_scatter(const SelParams* arr, const DevState* src_pinned, int count, hipStream_t s);
hipError_t launch_states_gather(const SelParams* arr, DevState* dst_pinned, int count, hipStream_t s);
hipError_t launch_group_iter(const SelParams* arr, int count, int dual, int max_nblk, int max_upd_blocks, hipStream_t s,
                             int maxR, int maxC);       // capacity of the largest node: sizes the dual select's LDS
hipError_t launch_group_init(const SelParams* arr, int count, hipStream_t s);
hipError_t launch_group_rhs_init(const SelParams* arr, int count, hipStream_t s);     // dual groups: contiguous RHS copy
hipError_t launch_rhs_init(const SelParams& p, hipStream_t s);
int update_blocks(int ld, int R);
struct BuildDesc { double* T; int32_t* basis; int32_t* shape; DevState* st; int ld, R, C, cut0; };
hipError_t launch_build_nodes(const double* T0, int ld0, int R0, int C0, const BuildDesc* descs, int count, int maxld, int maxR,
                              const int32_t* cvar, const double* ccoef, const double* czero, const double* crhs, hipStream_t s);
struct ChildDesc { const double* Tp; const int32_t* basis_p; double* T; int32_t* basis; int32_t* shape; DevState* st;
                   int ldp, Rp, Cp, ld, var, ik, is_ge, pad; double bound; };
hipError_t launch_build_children(const ChildDesc* descs, int count, int maxld, int maxR, hipStream_t s);
struct ParkDesc { const double* srcT; double* dstT; const int32_t* srcB; int32_t* dstB; size_t doubles; int m, pad; };
hipError_t launch_park_many(const ParkDesc* descs, int count, int blocks_per_node, hipStream_t s);
struct GatherDesc { const double* T; const int32_t* basis; int ld, R, C, off; };
int update_policy(int ld, int R);            // 0 = cache-resident update kernel, 1 = all-nt streaming, 2 = mixed-store streaming
hipError_t launch_gather_solution(const GatherDesc* descs, int count, double* out_rhs, int32_t* out_basis, hipStream_t s);
hipError_t launch_build_child(const double* Tp, int ldp, int Rp, int Cp, const int32_t* basis_p, double* T, int ld,
                              int var, int ik, int is_ge, double bound, int32_t* basis, hipStream_t s);
hipError_t launch_build_node(const double* T0, int ld0, int R0, int C0, double* T, int ld, int R, int C,
                             const int32_t* cvar, const double* ccoef, const double* czero, const double* crhs,
                             int32_t* basis, hipStream_t s);
hipError_t kernels_init();          // one-time function attributes
// FP64 matrix-core GEMM (lpx_mfma.hip): C = I - A*B (mode 0, max |C_ij| -> *absmax as double bits) or C = D + A*B (mode 1)
hipError_t launch_dgemm_mfma(const double* A, int lda, const double* B, int ldb, double* C, int ldc, const double* D, int ldd,
                             int M, int N, int K, int mode, unsigned long long* absmax, hipStream_t s);
// resident group loop (lpx_resident_group.hip): one entry per node of a launch
struct ResNode {
    double* T; int ld, R, C;
    int32_t* basis; int32_t* trace; int trace_cap;
    DevState* st;
    DevState* st_host;       // pinned host mirror of the fields the launch changes (the host reads it instead of copying `st` back)
    unsigned long long* xr;  // [2][mcap][2][2]  (a, rhs) granule pairs per row
    unsigned long long* xp;  // [2][ld+8][2]     pivot row granules, then the header {q}
    unsigned* xgen;
    int mcap, dual;
    double eps, tol_fdf, tol_dual, tol_primal;
    int max_iter, fdf_guard, cleanup;
};
size_t resident_group_lds(int R, int C, int ld, int grid);
hipError_t resident_group_init();
// register-resident variant (lpx_resident_regs.hip): node rows in VGPRs, more nodes per launch
hipError_t resident_regs_init();
int resident_regs_shape(int maxC, int min_ld, int mmax, int* rpw_max);   // 0 = does not fit; else the kernel configuration (1, 2, 3)
size_t resident_regs_lds(int R, int C, int rt, int cfg);
size_t resident_regs_lds_budget();                  // dynamic LDS a workgroup of the register-resident kernel may ask for
hipError_t launch_resnode_states_scatter(const void* nodes_dev, const DevState* src_pinned, int count, hipStream_t s);
hipError_t launch_resident_regs(const void* nodes_dev, int nodes, int grid, int cfg, int rt, size_t lds, int chunk, hipStream_t s);
hipError_t launch_resident_group(const void* nodes_dev, int nodes, int grid, size_t lds, int chunk, hipStream_t s);
// resident primal loop (lpx_resident.hip)
hipError_t resident_init();
int resident_plan(int R, int C, int ld, int* grid, int* rpw, size_t* lds);      // 0 = does not fit on chip
hipError_t launch_resident_primal(double* T, int ld, int R, int C, int grid, int rpw, size_t lds, int mcap,
                                  int32_t* basis, int32_t* trace, int trace_cap, DevState* st,
                                  unsigned long long* xr, unsigned long long* xp, unsigned* xgen,
                                  double eps, double tol, int max_iter, int chunk, hipStream_t s);

// resident primal loop with column-owning workgroups (lpx_resident_col.hip): one exchange per pivot
hipError_t resident_col_init();
int resident_col_plan(int R, int C, int* grid, int* cpw, size_t* lds);          // 0 = does not fit on chip
size_t resident_col_xc_bytes(int grid);
size_t resident_col_xq_bytes(int grid, int R);
hipError_t launch_resident_primal_col(double* T, int ld, int R, int C, int grid, int cpw, size_t lds,
                                      int32_t* basis, int32_t* trace, int trace_cap, DevState* st,
                                      unsigned long long* xc, unsigned long long* xq, unsigned* xgen,
                                      double eps, double tol, int max_iter, int chunk, hipStream_t s);

void set_error(const std::string& msg);
// Device memory the library keeps for reuse after its owner is gone (the chunk cache of destroyed parent stores, lpx_tableau.cpp):
// trim_device_caches() gives all of it back; malloc_retry() is hipMalloc that does so and tries once more before it reports
// hipErrorOutOfMemory -- every large allocation of the library goes through it.
void trim_device_caches();
hipError_t malloc_retry(void** p, size_t bytes);
int ensure_device();                // binds a device and sets kernel attributes once
double now_ms();
// hipStreamCreate costs 4-9 ms on this stack (an HSA queue each) while the hardware runs a handful of queues anyway:
// handles borrow one of a few process-wide streams (round robin) and never destroy them.
hipStream_t borrow_stream();

// Generic device-resident loop: enqueue `batch` iterations (eager, hipGraph replay, or event-
// bracketed), poll the device state once per batch, fire pivot callbacks from the trace.
struct LoopCtx {
    hipStream_t stream = nullptr;
    DevState* st = nullptr;            // device
    DevState* hst = nullptr;           // pinned host mirror
    int32_t* trace = nullptr; int trace_cap = 0;
    std::vector<hipEvent_t>* events = nullptr;
    hipGraphExec_t* gexec = nullptr;   // cached graph of *g_batch iterations keyed by *g_key
    int* g_batch = nullptr;
    std::string* g_key = nullptr;
    std::string key;                   // identifies the captured parameter set
    std::function<int(hipStream_t, hipEvent_t, hipEvent_t)> enqueue_iter;  // one iteration
    std::function<int(hipStream_t)> prologue;                               // once, before the loop
    int launches_per_iter = 2;
    bool profile_maps = true;          // pivot k of a batch == k-th enqueued iteration
    int start_iter = 0;                // pivots already done on this tableau by another path (resident loop hand-over)
};
// graph executables parked for `owner` (the address of a handle's gexec slot): destroyed with the handle, or when its buffers change
void graph_cache_drop_owner(const void* owner);
int run_device_loop(LoopCtx& c, const DevState& init, const lpx_run_opts* o, long long budget,
                    lpx_pivot_cb cb, void* user, lpx_stats* stats);

// Resumable form of the same loop: begin(); { submit(); complete(); } until done(); finish().
// submit() only enqueues (one batch + an async copy of the state to pinned memory), so several
// runs on different streams overlap on the device (lpx_multi_run).
class LoopRun {
public:
    LoopRun() = default;
    LoopRun(const LoopRun&) = delete;
    LoopRun& operator=(const LoopRun&) = delete;
    int begin(const LoopCtx& c, const DevState& init, const lpx_run_opts* o, long long budget,
              lpx_pivot_cb cb, void* user);
    int submit();
    int complete();
    bool done() const { return status_ != LPX_RUNNING || enq_ >= budget_; }
    bool may_submit() const { return enq_ < budget_; }
    int in_flight() const { return head_ - tail_; }
    int finish(lpx_stats* stats);          // returns the final status (or an error)
    ~LoopRun();
private:
    // up to two batches in flight: each submit() copies the state into its own pinned slot and records its own event, so
    // complete() waits for the OLDEST batch only while the next one is already queued behind it (run_device_loop)
    DevState* stage_ = nullptr; hipEvent_t ev_[2] = {nullptr, nullptr}; int head_ = 0, tail_ = 0;
    LoopCtx c_; lpx_run_opts o_{}; long long budget_ = 0, enq_ = 0;
    lpx_pivot_cb cb_ = nullptr; void* user_ = nullptr;
    lpx_stats local_{}; int batch_ = 64; bool graph_ = false;
    int fired_ = 0, iter_before_ = 0, status_ = LPX_RUNNING, init_phase_ = 2;
    double t0_ = 0;
};

}  // namespace lpx

#define LPX_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            lpx::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));             \
            return LPX_EDEVICE;                                                            \
        }                                                                                  \
    } while (0)
