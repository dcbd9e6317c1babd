// lpx_resident_group.hip -- several node LPs at once, each resident in the LDS of its own slice of the chip.
//
// Branch-and-bound nodes of config 4 are 8 MB tableaux: one fills a quarter of the chip's LDS.  This kernel is the
// resident loop of lpx_resident.hip generalised in two directions:
//   * blockIdx.y selects the node: `gridDim.x` workgroups (one per CU) own the rows of one tableau, and gridDim.y
//     such groups run side by side -- 4 nodes x 64 CUs instead of 32 nodes taking turns through HBM;
//   * the full state machine of the dual path (lpx_select_body in lpx_kernels.hip): ForceDualFeasibility
//     (Models/DualSimplex.cs:195-228), the dual loop (:36-113) and the repaired mode's primal clean-up, with the
//     primal loop (Models/PrimalSimplex.cs:92-124) as the special case "phase 2 from the start".
// Exchange 1 carries ONE value per row, chosen by the phase the next decision is in: the ratio rhs_i / T[i, q] for the
// entering column of the primal-like phases (every workgroup knows q from its objective replica; +inf when ineligible),
// or rhs_i for the dual loop's leaving row (most negative RHS, first index).  A phase hop that needs the other kind
// costs one exchange round without a pivot (once or twice per LP).  In the dual loop the entering column is chosen by the OWNER of row r -- it has
// the row and the objective replica, so the column ratio scan of :79-91 is local -- and travels as a header
// granule behind the normalised row.  Control state (phase, counters) is replicated: every workgroup of a group
// sees the same gathered data and takes the same decisions.
// Arithmetic per element is that of lpx_update / lpx_select, so results are bit-identical to the streaming kernels.
#include "lpx_resident.h"
#include <cstdlib>

namespace lpx {

struct ResGroupParams { const ResNode* nodes; int chunk; int mute; int defer; };   // mute: diagnostic, LPX_RESIDENT_TEST_MUTE; defer: see the round loop

#ifdef LPX_STAMPS
#define RG_T0 unsigned long long rg_prev_ = __builtin_amdgcn_s_memtime();
#define RG_T(slot) do { if (threadIdx.x == 0 && blockIdx.x == (gridDim.x > 1 ? 1 : 0) && blockIdx.y == 0) { unsigned long long n_ = __builtin_amdgcn_s_memtime(); P.xp[4 * ((size_t)P.ld + 8) + (slot)] += n_ - rg_prev_; rg_prev_ = n_; } } while (0)
#else
#define RG_T0
#define RG_T(slot) do {} while (0)
#endif

__device__ __forceinline__ int first4_first_min_below(const double* v, int L, double eps, double* s_v, int* s_i)
{
    MinIdx m; m.v = -eps; m.i = INT_MAX;
    if (threadIdx.x < RS_RT)
        for (int j = threadIdx.x; j < L; j += RS_RT) { const double x = v[j]; if (x < m.v) { m.v = x; m.i = j; } }
    m = first4_min_idx(m, s_v, s_i);
    __syncthreads();                                // s_v / s_i may be reused at once
    return m.i == INT_MAX ? -1 : m.i;
}

__global__ __launch_bounds__(RS_NT, 1) void lpx_resident_group(ResGroupParams GP)
{
    extern __shared__ __align__(16) double rs_lds[];
    __shared__ double s_v[RS_NT / 64];
    __shared__ int s_i[RS_NT / 64];
    __shared__ int s_out;

    const ResNode& N = GP.nodes[blockIdx.y];   // rarely used fields are read from the record where they are needed
    struct { double* T; int ld, R, C; unsigned long long* xr; unsigned long long* xp; int mcap, dual; } P;
    P.T = N.T; P.ld = N.ld; P.R = N.R; P.C = N.C; P.xr = N.xr; P.xp = N.xp; P.mcap = N.mcap; P.dual = N.dual;
    const double eps = N.eps;
    DevState* st = N.st;
    if (st->status != LPX_RUNNING) return;
    if ((GP.mute == 1 || (GP.mute == 2 && st->iter > 0)) && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) return;   // plays dead
    if (rs_abort_raised(st)) return;            // this node's launch is already lost (see lpx_resident.hip)
    if (GP.mute == 3 && blockIdx.x == gridDim.x - 1 && blockIdx.y == 0) rs_wait_for_abort(st);                      // a LATE workgroup
    const int t = threadIdx.x, w = blockIdx.x, G = gridDim.x;
    const int C = P.C, m = P.R - 1, rhsc = C - 1;
    const int rpw = (m + G - 1) / G;
    const int row0 = w * rpw;
    const int nloc = max(0, min(rpw, m - row0));
    const int mp = (m + 1) & ~1;
    const int gld = P.ld;                       // leading dimension in HBM (padded to 16 doubles)
    const int ld = (C + 1) & ~1;                // in LDS: just even, so that a fourth 8 MB node fits on the chip
    double* tile = rs_lds;                      // [rpw][ld]
    double* obj = tile + (size_t)rpw * ld;      // [ld]
    double* prow = obj + ld;                    // [ld]
    double* col = prow + ld;                    // [m]  gathered per-row values: ratios (phases 0 / 2) or rhs (phase 1)
    double* fac = col + mp;                     // [rpw+1]

    for (int i = 0; i < nloc; ++i) {
        const double* src = P.T + (size_t)(row0 + i) * gld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = *reinterpret_cast<const double2*>(src + j);
    }
    {
        const double* src = P.T + (size_t)m * gld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            *reinterpret_cast<double2*>(obj + j) = *reinterpret_cast<const double2*>(src + j);
            *reinterpret_cast<double2*>(prow + j) = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    int phase = P.dual ? st->phase : 2;
    int fdf_count = st->fdf_count, dual_iter = st->dual_iter, primal_count = st->primal_count, iter = st->iter;
    unsigned gen = *N.xgen;
    int status = LPX_RUNNING;
    bool hung = false;
    int r = -1, qlast = -1;
    // entering column the primal-like phases would take now (ChooseEntering, Models/PrimalSimplex.cs:205-220)
    int qc = (phase != 1) ? first4_first_min_below(obj, rhsc, eps, s_v, s_i) : -1;
    bool publish_now = GP.chunk > 0;            // rows' (a, rhs) for the first round of this launch

    // DEFERRED UPDATE (GP.defer, r03; as in lpx_resident_primal and lpx_resident_group_r): the rank-1 update of pivot k (factors `fac`,
    // row `prow`, its owner's row `pend_skip` already normalised in place) is applied in round k+1 between the decision and the
    // arrival of pivot k+1's row, where a workgroup used to sleep and poll.  Whoever reads the tile before that forms the value as the
    // update would (mul, then sub): the republish path its two columns, the owner of pivot k+1's row that row -- in place, ahead of the
    // rest, which then skips it.  The last pivot's update is applied behind the loop.
    const bool defer = GP.defer != 0;
    bool pend = false; int pend_skip = -1;
    RG_T0
    const int t_outer = t;
    for (int k = 0; k < GP.chunk; ++k) {
        // Opaque per-round copy of the lane index: without it the compiler hoists dozens of lane-dependent addresses
        // out of the round loop, runs out of the 128 VGPRs a 1024-lane workgroup gets and reloads them from scratch
        // on the critical path (the owner's publish loop waited ~1 us for three scratch loads).
        int t = t_outer;
        asm volatile("" : "+v"(t));
        if (publish_now) {
            // ForceDualFeasibility ends by itself when its guard is used up or no column is left (:202,:206-208): the hop
            // to the dual loop is taken here already, so that the rows publish what that loop needs (their rhs).
            if (phase == 0 && (fdf_count >= N.fdf_guard || qc < 0)) { phase = 1; qc = -1; }
            if (t < nloc) {
                double rhs0 = tile[(size_t)t * ld + rhsc];
                double a0 = (phase != 1 && qc >= 0) ? tile[(size_t)t * ld + qc] : 0.0;
                if (pend && t != pend_skip) {                                   // the two columns as the pending update will leave them
                    const double f = fac[t];
                    double prod = f * prow[rhsc]; rhs0 = rhs0 - prod;
                    if (phase != 1 && qc >= 0) { prod = f * prow[qc]; a0 = a0 - prod; }
                }
                double v = rhs0;
                if (phase != 1) v = a0 > eps ? rhs0 / a0 : __builtin_inf();     // :229-233
                rs_publish(P.xr + 2 * ((size_t)((gen + 1u) & 1u) * P.mcap + row0 + t), v, gen + 1u);
            }
            publish_now = false;
        }
        ++gen;
        const int par = (int)(gen & 1u);
        // ---- exchange 1: (a_i, rhs_i) of every row ------------------------------------------------------------
        int fail = 0;
        for (int base = t; base < m; base += RS_NT * RS_FETCH) {
            int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < m) cnt = u + 1; }
            if (!rs_gather(P.xr + 2 * (size_t)par * P.mcap, idx, cnt, gen, val)) fail = 1;
#pragma unroll
            for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) col[idx[u]] = val[u];
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        RG_T(1);

        // ---- the decision of lpx_select_body, replicated ------------------------------------------------------------
        int q = -1, final_status = LPX_RUNNING;
        bool republish = false;
        r = -1;
        for (int hop = 0; hop < 3 && final_status == LPX_RUNNING && r < 0 && !republish; ++hop) {
            if (phase == 1) {                                                   // dual loop, Models/DualSimplex.cs:36-113
                if (dual_iter >= N.max_iter) { final_status = LPX_ITER_LIMIT; break; }
                r = first4_first_min_below(col, m, eps, s_v, s_i);              // most negative RHS, first index (:45-55)
                if (r < 0) {
                    if (N.cleanup) {
                        const int qe = first4_first_min_below(obj, rhsc, eps, s_v, s_i);
                        if (qe >= 0) { phase = 2; qc = qe; republish = true; break; }   // the rows' ratios for qe are not out yet
                    }
                    final_status = LPX_OPTIMAL; break;
                }
                q = -2;                                                         // chosen by the owner of row r below
            } else {
                // ForceDualFeasibility (phase 0, :195-228) and the primal loop (phase 2, PrimalSimplex.cs:92-124)
                // (a hop 0 -> 1 needs the rows' rhs instead of their ratios: one exchange round without a pivot)
                if (phase == 0 && fdf_count >= N.fdf_guard) { phase = 1; qc = -1; republish = true; break; }
                if (phase == 2 && primal_count >= N.max_iter - dual_iter) { final_status = LPX_ITER_LIMIT; break; }
                q = qc;
                if (q < 0) { if (phase == 0) { phase = 1; qc = -1; republish = true; break; } final_status = LPX_OPTIMAL; break; }
                r = rs_hysteresis(m, phase == 0 ? N.tol_fdf : N.tol_primal, col, s_v, s_i, &s_out);
                if (r < 0) { q = -1; if (phase == 0) { phase = 1; qc = -1; republish = true; break; } final_status = LPX_UNBOUNDED; break; }
            }
        }
        if (republish) { publish_now = true; continue; }                        // one exchange round without a pivot
        if (final_status != LPX_RUNNING || r < 0) { status = (final_status == LPX_RUNNING) ? LPX_OPTIMAL : final_status; break; }

        RG_T(2);
        // ---- exchange 2: the owner normalises row r (and, in the dual loop, chooses the entering column) ---------
        const int owner = r / rpw, rl = r - owner * rpw;
        u64* xp = P.xp + 2 * (size_t)par * (gld + 8);
        if (w == owner) {
            double* prw = tile + (size_t)rl * ld;
            if (pend && rl != pend_skip) {                                      // the pending update of THIS row first, in place
                const double fp = fac[rl];
                for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
                    const double2 p = *reinterpret_cast<const double2*>(prow + j);
                    double2 v = *reinterpret_cast<double2*>(prw + j);
                    double prod = fp * p.x; v.x = v.x - prod;
                    prod = fp * p.y; v.y = v.y - prod;
                    *reinterpret_cast<double2*>(prw + j) = v;
                }
                __syncthreads();
            }
            if (phase == 1) {                                                   // entering column of the dual loop, :79-91
                if ((t >> 6) == 0) {
                    const int win = rs_exact_col_scan(rhsc, N.tol_dual, prw, obj, eps);
                    if (t == 0) s_out = win;
                }
                __syncthreads();
                q = s_out;
            }
            if (q >= 0) {
                const double piv = prw[q];
                __syncthreads();
                for (int j = t; j < C; j += RS_NT) {
                    const double p = prw[j] / piv;
                    rs_publish(xp + 2 * (size_t)j, p, gen);
                    prw[j] = p;
                    if (!defer) prow[j] = p;                                    // defer: prow is still the pending update's row
                }
            }
            if (phase == 1) {
                __syncthreads();
                if (t == 0) rs_publish(xp + 2 * (size_t)gld, (double)q, gen);   // header {q} behind the row
            }
        }
        const bool had_pending = pend;
        if (defer) {
            if (pend) {
                const int skip2 = (w == owner) ? rl : -1;
                for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
                    const double2 p = *reinterpret_cast<const double2*>(prow + j);
                    for (int i = 0; i < nloc; ++i) {
                        if (i == pend_skip || i == skip2) continue;
                        const double f = fac[i];
                        double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                        double prod = f * p.x; v.x = v.x - prod;
                        prod = f * p.y; v.y = v.y - prod;
                        *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
                    }
                }
                pend = false;
            }
            rs_barrier_lds();                   // prow may be rewritten now
            if (w == owner && q >= 0) { const double* prw = tile + (size_t)rl * ld; for (int j = t; j < C; j += RS_NT) prow[j] = prw[j]; }
        }
        if (w == owner) {
        } else if (phase != 1) {
            // q is known (it came from the objective replica): exactly the consumer side of lpx_resident_primal --
            // sleep through the owner's divide + store, one look at the row, else poll one granule (the last column).
            if (!had_pending) {
                __builtin_amdgcn_s_sleep(15);
                for (int z = 0; z < C; z += RS_NT) __builtin_amdgcn_s_sleep(5);
            }
            bool first = true;
            for (int base = t; base < C; base += RS_NT * RS_FETCH) {
                int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < C) cnt = u + 1; }
                unsigned pend = (1u << cnt) - 1u;
                if (first && !rs_gather(xp, idx, cnt, gen, val, 1u, &pend)) {
                    if (!rs_wait(xp + 2 * (size_t)(C - 1), gen)) fail = 1;
                }
                first = false;
                if (pend && !rs_gather(xp, idx, cnt, gen, val, RS_SPIN_MAX, &pend)) fail = 1;
#pragma unroll
                for (int u = 0; u < RS_FETCH; ++u) if (u < cnt) prow[idx[u]] = val[u];
            }
        } else {
            // sleep through the owner's work, then ONE look at the header and the row together; if they are not all
            // there yet, wait for the header alone (one granule) and gather what is missing.
            if (!had_pending) {
                __builtin_amdgcn_s_sleep(15);
                for (int z = 0; z < C; z += RS_NT) __builtin_amdgcn_s_sleep(5);
                if (phase == 1) __builtin_amdgcn_s_sleep(25);                   // the owner scans its row first
            }
            bool first = true;
            double hq = 0.0; unsigned hpend = 1u;
            for (int base = t; first || base < C; base += RS_NT * 3) {       // every lane fetches the header at least
                int idx[RS_FETCH]; double val[RS_FETCH]; int cnt = 0;
#pragma unroll
                for (int u = 0; u < 3; ++u) { idx[u] = base + u * RS_NT; if (idx[u] < C) cnt = u + 1; }
                unsigned pend = (1u << cnt) - 1u;
                if (first) {
                    idx[cnt] = gld;                                             // the header rides in the spare slot
                    unsigned p2 = pend | (1u << cnt);
                    rs_gather(xp, idx, cnt + 1, gen, val, 1u, &p2);
                    if (!((p2 >> cnt) & 1u)) { hq = val[cnt]; hpend = 0u; }
                    pend = p2 & ((1u << cnt) - 1u);
                    if (hpend) {
                        int hidx[RS_FETCH] = {gld, gld, gld, gld}; double hval[RS_FETCH];
                        if (!rs_gather(xp, hidx, 1, gen, hval)) fail = 1;
                        hq = hval[0]; hpend = 0u;
                    }
                    q = fail ? -1 : (int)hq;
                    first = false;
                }
                if (q < 0) break;
                if (pend && !rs_gather(xp, idx, cnt, gen, val, RS_SPIN_MAX, &pend)) fail = 1;
#pragma unroll
                for (int u = 0; u < 3; ++u) if (u < cnt) prow[idx[u]] = val[u];
            }
        }
        if (__syncthreads_or(fail)) { hung = true; break; }
        if (q < 0) { r = -1; status = LPX_INFEASIBLE; break; }                  // :92-96 (dual loop only)
        RG_T(3);

        // ---- column factors of this pivot, objective replica, next entering column ------------------------------
        if (t < nloc) fac[t] = tile[(size_t)t * ld + q];
        if (t == RS_NT - 1) fac[rpw] = obj[q];
        __syncthreads();
        const double fobj = fac[rpw];
        const int skip = (w == owner) ? rl : -1;
        MinIdx best; best.v = -eps; best.i = INT_MAX;
        if (t < RS_RT) {
            for (int j = 2 * t; j < ld; j += 2 * RS_RT) {
                const double2 p = *reinterpret_cast<const double2*>(prow + j);
                double2 o = *reinterpret_cast<double2*>(obj + j);
                double prod = fobj * p.x; o.x = o.x - prod;
                prod = fobj * p.y; o.y = o.y - prod;
                *reinterpret_cast<double2*>(obj + j) = o;
                if (j < rhsc && o.x < best.v) { best.v = o.x; best.i = j; }
                if (j + 1 < rhsc && o.y < best.v) { best.v = o.y; best.i = j + 1; }
            }
        }
        best = first4_min_idx(best, s_v, s_i);
        if (w == 0 && t == 0) {
            N.basis[r] = q;                                                     // basis[leaving] = entering, :110
            if (iter < N.trace_cap) { N.trace[2 * iter] = r; N.trace[2 * iter + 1] = q; }
        }
        qlast = q;
        ++iter;
        if (phase == 0) ++fdf_count; else if (phase == 1) ++dual_iter; else ++primal_count;
        qc = (phase != 1) ? (best.i == INT_MAX ? -1 : best.i) : -1;
        RG_T(4);
        // ---- lookahead: next round's (a, rhs) leave before the bulk of the update ---------------------------------
        if (k + 1 < GP.chunk) {
            // the hop out of ForceDualFeasibility that is already certain (:202,:206-208) is taken before publishing
            if (phase == 0 && (fdf_count >= N.fdf_guard || qc < 0)) { phase = 1; qc = -1; }
            if (t < nloc) {
                double a = 0.0, rhs;
                if (t == skip) { if (qc >= 0) a = prow[qc]; rhs = prow[rhsc]; }
                else {
                    const double f = fac[t];
                    if (qc >= 0) { const double prod = f * prow[qc]; a = tile[(size_t)t * ld + qc] - prod; }
                    const double prod2 = f * prow[rhsc]; rhs = tile[(size_t)t * ld + rhsc] - prod2;
                }
                const double v = (phase == 1) ? rhs : (a > eps ? rhs / a : __builtin_inf());
                rs_publish(P.xr + 2 * ((size_t)(par ^ 1) * P.mcap + row0 + t), v, gen + 1u);
            }
        }
        __syncthreads();                        // the lookahead read columns qc and rhs before anyone rewrites them
        RG_T(0);
        // ---- rank-1 update of the local rows, :250-256 -------------------------------------------------------
        if (defer) { pend = true; pend_skip = skip; }
        else
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            const double2 p = *reinterpret_cast<const double2*>(prow + j);
            for (int i = 0; i < nloc; ++i) {
                if (i == skip) continue;
                const double f = fac[i];
                double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                double prod = f * p.x; v.x = v.x - prod;
                prod = f * p.y; v.y = v.y - prod;
                *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
            }
        }
        __syncthreads();
        RG_T(5);
    }

    if (hung) {
        if (t == 0) { atomicOr(&st->pad[1], 1); if (N.st_host) N.st_host->pad[1] = 1; }
        return;
    }
    if (pend) {                                 // the last pivot's update, on the way out
        for (int j = 2 * t; j < ld; j += 2 * RS_NT) {
            const double2 p = *reinterpret_cast<const double2*>(prow + j);
            for (int i = 0; i < nloc; ++i) {
                if (i == pend_skip) continue;
                const double f = fac[i];
                double2 v = *reinterpret_cast<double2*>(tile + (size_t)i * ld + j);
                double prod = f * p.x; v.x = v.x - prod;
                prod = f * p.y; v.y = v.y - prod;
                *reinterpret_cast<double2*>(tile + (size_t)i * ld + j) = v;
            }
        }
        __syncthreads();
    }
    for (int i = 0; i < nloc; ++i) {
        double* dst = P.T + (size_t)(row0 + i) * gld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(tile + (size_t)i * ld + j);
    }
    if (w == 0) {
        double* dst = P.T + (size_t)m * gld;
        for (int j = 2 * t; j < ld; j += 2 * RS_NT)
            *reinterpret_cast<double2*>(dst + j) = *reinterpret_cast<const double2*>(obj + j);
        if (t == 0) {
            st->status = status; st->iter = iter; st->phase = phase;
            st->fdf_count = fdf_count; st->dual_iter = dual_iter; st->primal_count = primal_count;
            st->r = status == LPX_RUNNING ? r : -1; st->q = status == LPX_RUNNING ? qlast : -1;
            if (DevState* hm = N.st_host) {              // the same fields, straight into the host's pinned copy
                hm->status = status; hm->iter = iter; hm->phase = phase;
                hm->fdf_count = fdf_count; hm->dual_iter = dual_iter; hm->primal_count = primal_count;
                hm->r = status == LPX_RUNNING ? r : -1; hm->q = status == LPX_RUNNING ? qlast : -1;
            }
            *N.xgen = gen;
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------
size_t resident_group_lds(int R, int C, int ld, int grid)
{
    const int m = R - 1;
    const int rpw = (m + grid - 1) / grid;
    (void)ld;                                   // LDS rows are only as wide as the tableau (even), not the HBM pitch
    return sizeof(double) * ((size_t)(rpw + 2) * ((C + 1) & ~1) + (size_t)((m + 1) & ~1) + (size_t)rpw + 2);
}

hipError_t resident_group_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lpx_resident_group),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
}

// every node's initial state record in one launch: src is the host's pinned array, node i <-> src[i]
__global__ __launch_bounds__(64) void lpx_resnode_states_scatter(const ResNode* __restrict__ nodes, const DevState* __restrict__ src)
{
    const int32_t* s = reinterpret_cast<const int32_t*>(src + blockIdx.x);
    int32_t* d = reinterpret_cast<int32_t*>(nodes[blockIdx.x].st);
    for (int k = threadIdx.x; k < (int)(sizeof(DevState) / sizeof(int32_t)); k += 64) d[k] = s[k];
}
hipError_t launch_resnode_states_scatter(const void* nodes_dev, const DevState* src_pinned, int count, hipStream_t s)
{
    hipLaunchKernelGGL(lpx_resnode_states_scatter, dim3(count), dim3(64), 0, s, static_cast<const ResNode*>(nodes_dev), src_pinned);
    return hipGetLastError();
}

hipError_t launch_resident_group(const void* nodes_dev, int nodes, int grid, size_t lds, int chunk, hipStream_t s)
{
    ResGroupParams p; p.nodes = static_cast<const ResNode*>(nodes_dev); p.chunk = chunk;
    static const int mute = [] { const char* e = std::getenv("LPX_RESIDENT_TEST_MUTE"); return e ? std::atoi(e) : 0; }();
    p.mute = mute;
    static const int defer = [] { const char* e = std::getenv("LPX_RESIDENT_DEFER"); return (e && e[0] == '0') ? 0 : 1; }();   // diagnostic: 0 = update at the end of its own round
    p.defer = defer;
    hipLaunchKernelGGL(lpx_resident_group, dim3(grid, nodes), dim3(RS_NT), lds, s, p);
    return hipGetLastError();
}

}  // namespace lpx
