// lpx_knapsack.hip -- batched greedy fractional bounds for the 0/1 knapsack branch and bound
// (ComputeRelaxation, Models/BranchAndBoundKnapsack.cs:431-491).  One workgroup per B&B node.
//
// Items live in HBM in the reference's ratio order (OrderByDescending(Ratio).ThenByDescending(Profit),
// stable, :75-79).  A node is its list of fixed decisions (original index ascending, value 0/1) -- the
// reference's int[n] `Assigned` (:25) is 400 KB per node at n = 100 000 and is never shipped.
// Per node:  fixed-1 items are summed first in index order by one lane (the reference's order, :442-452);
// every lane then owns a contiguous slice of the sorted items, sums its undecided weights/profits,
// an exclusive scan gives each slice its starting totals, each lane replays the reference's test
// `weight + w_i <= cap + EPS` over its slice and the first failing sorted position in the block is the
// break item (:468-487).  Sums are exact (order-free) for integer data below 2^53, which is what the
// synthetic configuration uses; for other data the scan reorders additions (documented: 1e-9 relative).
#include "lpx_block.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace lpx {

static constexpr int KN_NT = 512;
static constexpr int KN_NW = KN_NT / 64;
static constexpr double KEPS = 1e-9;      // Models/BranchAndBoundKnapsack.cs:56

struct KnParams {
    int n; double cap;
    const double* ws; const double* ps;     // [n] weights / profits in ratio order
    const int32_t* pos;                     // [n] original index -> sorted position
    const double* w0; const double* p0;     // [n] weights / profits by original index
    int count;
    const int32_t* off; const int32_t* fidx; const int8_t* fval;
    double* out_profit; double* out_weight; int32_t* out_frac; double* out_fracval;
};

__device__ __forceinline__ double wave_incl_scan_sum(double x)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        double y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

__global__ __launch_bounds__(KN_NT) void knap_relax_batch(KnParams P)
{
    extern __shared__ unsigned int s_bits[];             // bitmap over sorted positions: 1 = fixed
    __shared__ double s_w[KN_NW], s_p[KN_NW];
    __shared__ double s_w1, s_p1;
    __shared__ int s_min[KN_NW];
    const int node = blockIdx.x;
    if (node >= P.count) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int n = P.n;
    const int nwords = (n + 31) >> 5;
    for (int k = t; k < nwords; k += KN_NT) s_bits[k] = 0u;
    __syncthreads();
    const int f0 = P.off[node], f1 = P.off[node + 1];
    for (int e = f0 + t; e < f1; e += KN_NT) {
        const int sp = P.pos[P.fidx[e]];
        atomicOr(&s_bits[sp >> 5], 1u << (sp & 31));
    }
    if (t == 0) {                                        // :442-452, original index order
        double w = 0.0, p = 0.0;
        for (int e = f0; e < f1; ++e)
            if (P.fval[e] == 1) { const int i = P.fidx[e]; w += P.w0[i]; p += P.p0[i]; }
        s_w1 = w; s_p1 = p;
    }
    __syncthreads();
    const double W1 = s_w1, P1 = s_p1;
    if (W1 > P.cap + KEPS) {                             // :455-456
        if (t == 0) { P.out_profit[node] = P1; P.out_weight[node] = W1; P.out_frac[node] = -1; P.out_fracval[node] = 0.0; }
        return;
    }
    const int per = (n + KN_NT - 1) / KN_NT;
    const int lo = t * per, hi = min(n, lo + per);
    double lw = 0.0, lp = 0.0;
    for (int s = lo; s < hi; ++s)
        if (!((s_bits[s >> 5] >> (s & 31)) & 1u)) { lw += P.ws[s]; lp += P.ps[s]; }
    // exclusive scan of (lw, lp) over lanes in order
    double iw = wave_incl_scan_sum(lw), ip = wave_incl_scan_sum(lp);
    if (lane == 63) { s_w[wave] = iw; s_p[wave] = ip; }
    __syncthreads();
    double bw = 0.0, bp = 0.0;
    for (int w = 0; w < wave; ++w) { bw += s_w[w]; bp += s_p[w]; }
    double accw = W1 + (bw + (iw - lw));
    double accp = P1 + (bp + (ip - lp));
    // replay the greedy test over the own slice
    int fail = INT_MAX; double fw = 0.0, fp = 0.0;       // totals just before the failing item
    for (int s = lo; s < hi; ++s) {
        if ((s_bits[s >> 5] >> (s & 31)) & 1u) continue;
        const double wi = P.ws[s];
        if (accw + wi <= P.cap + KEPS) { accw += wi; accp += P.ps[s]; }
        else { fail = s; fw = accw; fp = accp; break; }
    }
    int mn = fail;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mn = min(mn, __shfl_xor(mn, d, 64));
    if (lane == 0) s_min[wave] = mn;
    __syncthreads();
    int gmin = INT_MAX;
    for (int w = 0; w < KN_NW; ++w) gmin = min(gmin, s_min[w]);
    if (gmin == INT_MAX) {
        if (t == KN_NT - 1) {                            // everything fits: last lane holds the totals
            P.out_profit[node] = accp; P.out_weight[node] = accw; P.out_frac[node] = -1; P.out_fracval[node] = 0.0;
        }
    } else if (fail == gmin) {                           // owner of the break item (:474-487)
        double w = fw, p = fp; int frac = -1; double fv = 0.0;
        const double wi = P.ws[gmin];
        const double remain = P.cap - w;
        if (remain > KEPS && wi > KEPS) {
            fv = remain / wi;
            p += P.ps[gmin] * fv;
            w += wi * fv;
            frac = gmin;
        }
        P.out_profit[node] = p; P.out_weight[node] = w; P.out_frac[node] = frac; P.out_fracval[node] = fv;
    }
}

// ------------------------------------------------------------------------------------------------
// Prefix-sum form (all weights >= 0): one WAVE per node, O(depth * log n) instead of O(n).
// PW[s] / PP[s] are the running sums of weights / profits over the ratio-ordered items [0, s), built
// once on the host with the reference's left-to-right additions.  For a node with fixed set F the
// undecided weight in front of position t is U(t) = PW[t] - sum_{f in F, pos_f < t} w_f, non-decreasing
// in t, so the reference's greedy (`weight + w_i <= cap + EPS` until the first failure, :468-487) stops
// at j = t* - 1 with t* the smallest t in [1, n] such that W1 + U(t) > cap + EPS -- a binary search
// whose every probe is one wave reduction over the node's fixed entries.  j is undecided by minimality
// (a fixed j would give U(t*) == U(t* - 1)).  Sums are exact and order-free for integer data below 2^53
// (the synthetic configuration); other data is reproduced to 1e-9 relative only, as the scan kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double x)
{
    x += dpp_f64<0x111, 0xf>(0.0, x);
    x += dpp_f64<0x112, 0xf>(0.0, x);
    x += dpp_f64<0x114, 0xf>(0.0, x);
    x += dpp_f64<0x118, 0xf>(0.0, x);
    x += dpp_f64<0x142, 0xa>(0.0, x);
    x += dpp_f64<0x143, 0xc>(0.0, x);
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
    return __hiloint2double(hi, lo);
}

struct KnPrefix { const double* PW; const double* PP; };

static constexpr int KP_CACHE = 4;        // fixed entries cached in registers per lane (depth <= 256)

// DEPTH2: after the node itself (slot 3k) the wave also evaluates the node's two children, i.e. the node with its
// fractional item additionally fixed to 0 (slot 3k+1) and to 1 (slot 3k+2) -- the very relaxations the best-first
// loop asks for when it pops this node next, so the host finds them cached and launches half as often.  The extra
// decision is one more fixed entry (its ratio position is the node's `frac`, weight / profit from the sorted arrays).
template <bool DEPTH2>
__global__ __launch_bounds__(256) void knap_relax_prefix(KnParams P, KnPrefix X)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int node = blockIdx.x * 4 + wave;
    if (node >= P.count) return;
    const int n = P.n;
    const int f0 = P.off[node], d = P.off[node + 1] - f0;
    int cpos[KP_CACHE]; double cw[KP_CACHE], cp[KP_CACHE];
    double w1 = 0.0, p1 = 0.0;
#pragma unroll
    for (int k = 0; k < KP_CACHE; ++k) {
        const int e = lane + 64 * k;
        cpos[k] = INT_MAX; cw[k] = 0.0; cp[k] = 0.0;
        if (e < d) {
            const int i = P.fidx[f0 + e];
            cpos[k] = P.pos[i]; cw[k] = P.w0[i]; cp[k] = P.p0[i];
            if (P.fval[f0 + e] == 1) { w1 += cw[k]; p1 += cp[k]; }
        }
    }
    for (int e = lane + 64 * KP_CACHE; e < d; e += 64) {            // deeper nodes: the tail stays in memory
        const int i = P.fidx[f0 + e];
        if (P.fval[f0 + e] == 1) { w1 += P.w0[i]; p1 += P.p0[i]; }
    }
    const double W1n = wave_sum_f64(w1), P1n = wave_sum_f64(p1);   // :442-452 (exact for integer data)

    // one relaxation: the node's fixed list plus (optionally) the item at ratio position xpos fixed to xval
    auto solve = [&](int xpos, int xval, double xw, double xp, int slot) {
        const double W1 = W1n + (xval == 1 ? xw : 0.0), P1 = P1n + (xval == 1 ? xp : 0.0);
        if (W1 > P.cap + KEPS) {                                        // :455-456
            if (lane == 0) { P.out_profit[slot] = P1; P.out_weight[slot] = W1; P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0; }
            return -1;
        }
        // fixed weight / profit in front of position t
        auto fixed_before = [&](int t, double& fw, double& fp) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < KP_CACHE; ++k) if (cpos[k] < t) { a += cw[k]; b += cp[k]; }
            for (int e = lane + 64 * KP_CACHE; e < d; e += 64) {
                const int i = P.fidx[f0 + e];
                if (P.pos[i] < t) { a += P.w0[i]; b += P.p0[i]; }
            }
            fw = wave_sum_f64(a); fp = wave_sum_f64(b);
            if (xpos < t) { fw += xw; fp += xp; }
        };
        int lo = 1, hi = n + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            double fw, fp;
            fixed_before(mid, fw, fp);
            if (W1 + (X.PW[mid] - fw) > P.cap + KEPS) hi = mid; else lo = mid + 1;
        }
        double fw, fp;
        if (lo == n + 1) {                                              // everything undecided fits
            fixed_before(n, fw, fp);
            if (lane == 0) {
                P.out_profit[slot] = P1 + (X.PP[n] - fp); P.out_weight[slot] = W1 + (X.PW[n] - fw);
                P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0;
            }
            return -1;
        }
        const int j = lo - 1;                                           // first undecided item that does not fit
        fixed_before(j, fw, fp);
        double w = W1 + (X.PW[j] - fw), p = P1 + (X.PP[j] - fp);
        int frac = -1; double fv = 0.0;
        const double wi = P.ws[j];
        const double remain = P.cap - w;
        if (remain > KEPS && wi > KEPS) {                               // :476-484
            fv = remain / wi;
            p += P.ps[j] * fv;
            w += wi * fv;
            frac = j;
        }
        if (lane == 0) { P.out_profit[slot] = p; P.out_weight[slot] = w; P.out_frac[slot] = frac; P.out_fracval[slot] = fv; }
        return frac;
    };

    if (!DEPTH2) { solve(INT_MAX, 0, 0.0, 0.0, node); return; }
    const int frac = solve(INT_MAX, 0, 0.0, 0.0, 3 * node);
    if (frac < 0) {                                                     // no fractional item: nothing to branch on
        if (lane == 0) { P.out_frac[3 * node + 1] = -2; P.out_frac[3 * node + 2] = -2; }
        return;
    }
    const double xw = P.ws[frac], xp = P.ps[frac];
    solve(frac, 0, xw, xp, 3 * node + 1);
    solve(frac, 1, xw, xp, 3 * node + 2);
}

// ------------------------------------------------------------------------------------------------
// Device-resident node store.  The best-first loop creates every node as "its parent plus one decision"
// (Models/BranchAndBoundKnapsack.cs:207-209,:267-269), so a node's fixed list never has to travel: the parent's list
// already sits in HBM, the kernel writes the child's list next to it and evaluates the child -- and the child's own two
// children, as knap_relax_prefix<true> does -- from the parent's entries plus the new decision(s) as extras.  A job is
// 48 bytes instead of the whole list (100-250 entries per node at config 5), and the host keeps a heap of ids.
// A stored list is ordered by RATIO RANK (the position in the sorted arrays), one word per decision: rank | value << 31.
// In that order the fixed weight in front of a position t is a prefix of the list, which is what the wide search needs;
// the host turns ranks back into item indices (lpx_knapsack_node_list).
// ------------------------------------------------------------------------------------------------
struct KnJob { const uint32_t* parent; uint32_t* child; uint32_t* g0; uint32_t* g1; int32_t depth; int32_t item; int32_t val; int32_t pad; };

// Deep nodes (more than KW_CAP decisions): one binary search per relaxation, every probe one wave reduction over the
// parent's entries.  17 dependent probes per relaxation at n = 100 000 -- the form the wide kernel below replaces.
__global__ __launch_bounds__(256) void knap_expand(KnParams P, KnPrefix X, const KnJob* __restrict__ jobs)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wave;
    if (job >= P.count) return;
    const KnJob J = jobs[job];
    const int n = P.n, d = J.depth;
    const uint32_t* __restrict__ par = J.parent;
    // the new decision
    const int x1pos = P.pos[J.item]; const double x1w = P.ws[x1pos], x1p = P.ps[x1pos]; const int x1val = J.val;
    int cpos[KP_CACHE]; double cw[KP_CACHE], cp[KP_CACHE];
    double w1 = 0.0, p1 = 0.0;
    int before = 0;                                     // parent entries with a smaller rank (insertion point)
#pragma unroll
    for (int k = 0; k < KP_CACHE; ++k) {
        const int e = lane + 64 * k;
        cpos[k] = INT_MAX; cw[k] = 0.0; cp[k] = 0.0;
        if (e < d) {
            const uint32_t v = par[e];
            const int r = (int)(v & 0x7fffffffu);
            cpos[k] = r; cw[k] = P.ws[r]; cp[k] = P.ps[r];
            if (v >> 31) { w1 += cw[k]; p1 += cp[k]; }
            before += (r < x1pos) ? 1 : 0;
            J.child[e + ((r > x1pos) ? 1 : 0)] = v;
        }
    }
    for (int e = lane + 64 * KP_CACHE; e < d; e += 64) {            // the tail stays in memory
        const uint32_t v = par[e];
        const int r = (int)(v & 0x7fffffffu);
        if (v >> 31) { w1 += P.ws[r]; p1 += P.ps[r]; }
        before += (r < x1pos) ? 1 : 0;
        J.child[e + ((r > x1pos) ? 1 : 0)] = v;
    }
    before = (int)wave_sum_f64((double)before);
    if (lane == 0) J.child[before] = (uint32_t)x1pos | ((uint32_t)x1val << 31);
    const double W1n = wave_sum_f64(w1) + (x1val == 1 ? x1w : 0.0), P1n = wave_sum_f64(p1) + (x1val == 1 ? x1p : 0.0);

    auto solve = [&](int xpos, int xval, double xw, double xp, int slot) {
        const double W1 = W1n + (xval == 1 ? xw : 0.0), P1 = P1n + (xval == 1 ? xp : 0.0);
        if (W1 > P.cap + KEPS) {                                        // :455-456
            if (lane == 0) { P.out_profit[slot] = P1; P.out_weight[slot] = W1; P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0; }
            return -1;
        }
        auto fixed_before = [&](int t, double& fw, double& fp) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int k = 0; k < KP_CACHE; ++k) if (cpos[k] < t) { a += cw[k]; b += cp[k]; }
            for (int e = lane + 64 * KP_CACHE; e < d; e += 64) {
                const int r = (int)(par[e] & 0x7fffffffu);
                if (r < t) { a += P.ws[r]; b += P.ps[r]; }
            }
            fw = wave_sum_f64(a); fp = wave_sum_f64(b);
            if (x1pos < t) { fw += x1w; fp += x1p; }
            if (xpos < t) { fw += xw; fp += xp; }
        };
        int lo = 1, hi = n + 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            double fw, fp;
            fixed_before(mid, fw, fp);
            if (W1 + (X.PW[mid] - fw) > P.cap + KEPS) hi = mid; else lo = mid + 1;
        }
        double fw, fp;
        if (lo == n + 1) {
            fixed_before(n, fw, fp);
            if (lane == 0) {
                P.out_profit[slot] = P1 + (X.PP[n] - fp); P.out_weight[slot] = W1 + (X.PW[n] - fw);
                P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0;
            }
            return -1;
        }
        const int j = lo - 1;
        fixed_before(j, fw, fp);
        double w = W1 + (X.PW[j] - fw), p = P1 + (X.PP[j] - fp);
        int frac = -1; double fv = 0.0;
        const double wi = P.ws[j];
        const double remain = P.cap - w;
        if (remain > KEPS && wi > KEPS) {                               // :476-484
            fv = remain / wi;
            p += P.ps[j] * fv;
            w += wi * fv;
            frac = j;
        }
        if (lane == 0) { P.out_profit[slot] = p; P.out_weight[slot] = w; P.out_frac[slot] = frac; P.out_fracval[slot] = fv; }
        return frac;
    };
    const int frac = solve(INT_MAX, 0, 0.0, 0.0, 3 * job);
    if (frac < 0) {
        if (lane == 0) { P.out_frac[3 * job + 1] = -2; P.out_frac[3 * job + 2] = -2; }
        return;
    }
    const double xw = P.ws[frac], xp = P.ps[frac];
    solve(frac, 0, xw, xp, 3 * job + 1);
    solve(frac, 1, xw, xp, 3 * job + 2);
    // the lists of those two children (the child's list plus its fractional item, value 0 / 1): every relaxation the host
    // caches has its node in the store, so the host never has to send a list
    const int x2 = frac;
    int before2 = 0;
    for (int e = lane; e < d; e += 64) {
        const uint32_t v = par[e];
        const int r = (int)(v & 0x7fffffffu);
        before2 += (r < x2) ? 1 : 0;
        const int dst = e + ((r > x1pos) ? 1 : 0) + ((r > x2) ? 1 : 0);
        J.g0[dst] = v; J.g1[dst] = v;
    }
    before2 = (int)wave_sum_f64((double)before2);
    if (lane == 0) {
        const int q1 = before + ((x1pos > x2) ? 1 : 0), q2 = before2 + ((x2 > x1pos) ? 1 : 0);
        const uint32_t e1 = (uint32_t)x1pos | ((uint32_t)x1val << 31);
        J.g0[q1] = e1; J.g1[q1] = e1;
        J.g0[q2] = (uint32_t)x2; J.g1[q2] = (uint32_t)x2 | 0x80000000u;
    }
}

// The wide form (nodes of at most KW_CAP decisions -- every node of config 5).  The break position t* is the smallest t
// with  g(t) = W1 + PW[t] - fixed_before(t) > cap + EPS,  g monotone in t.  Instead of halving [1, n] with one probe per
// step (17 dependent memory round trips per relaxation, 51 per job: the 20 us the r02 trace shows for this kernel), the
// 64 lanes probe 64 positions at once:
//   * the parent's list is ordered by rank, so fixed_before(t) = prefix[#entries with rank < t]: each lane holds eight
//     consecutive entries, one wave scan gives the exclusive prefixes, ranks + prefixes go to LDS (10 KB per wave) and a
//     lane answers its own t by a binary search there (<= 9 LDS reads, no memory traffic);
//   * a round = one gather PW[t_lane] + one ballot; [1, n + 1] shrinks by 64x per round (three rounds at n = 100 000),
//     and the first round is aimed at where the answer almost always is: within +-32 positions of the branching item
//     (fixing the parent's fractional item to 0 / 1 moves the break by about one undecided item).
// Same predicate, same t*, same closing arithmetic as the deep form above: identical outputs for integer data (sums
// exact below 2^53), 1e-9 relative otherwise, as for the host-list kernels.
static constexpr int KW_PER = 8;
static constexpr int KW_CAP = 64 * KW_PER;
struct KwWave { double fw[KW_CAP + 1]; double fp[KW_CAP + 1]; int rank[KW_CAP + 1]; int pad; };

__global__ __launch_bounds__(256) void knap_expand_w(KnParams P, KnPrefix X, const KnJob* __restrict__ jobs)
{
    __shared__ KwWave lds[4];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wave;
    const bool live = job < P.count;
    KwWave& L = lds[wave];
    const KnJob J = jobs[live ? job : 0];
    const int n = P.n, d = live ? J.depth : 0;
    const uint32_t* __restrict__ par = J.parent;
    const int x1pos = P.pos[J.item]; const double x1w = P.ws[x1pos], x1p = P.ps[x1pos]; const int x1val = J.val;

    uint32_t ev[KW_PER]; int rk[KW_PER]; double wk[KW_PER], pk[KW_PER];
#pragma unroll
    for (int h = 0; h < KW_PER / 4; ++h) {
        const int e0 = lane * KW_PER + 4 * h;
        uint4 q = {0u, 0u, 0u, 0u};
        if (e0 < d) q = *reinterpret_cast<const uint4*>(par + e0);        // lists are carved in 16-byte granules
        ev[4 * h + 0] = q.x; ev[4 * h + 1] = q.y; ev[4 * h + 2] = q.z; ev[4 * h + 3] = q.w;
    }
    double lw = 0.0, lp = 0.0, w1 = 0.0, p1 = 0.0;
#pragma unroll
    for (int k = 0; k < KW_PER; ++k) {
        const int e = lane * KW_PER + k;
        const bool valid = e < d;
        const int r = valid ? (int)(ev[k] & 0x7fffffffu) : INT_MAX;
        rk[k] = r;
        wk[k] = valid ? P.ws[r] : 0.0; pk[k] = valid ? P.ps[r] : 0.0;
        lw += wk[k]; lp += pk[k];
        if (valid && (ev[k] >> 31)) { w1 += wk[k]; p1 += pk[k]; }
        if (valid && live) J.child[e + ((r > x1pos) ? 1 : 0)] = ev[k];
    }
    {   // exclusive prefixes over the lanes, then per entry
        double sw = wave_incl_scan_sum(lw), sp = wave_incl_scan_sum(lp);
        double rw = __shfl_up(sw, 1, 64), rp = __shfl_up(sp, 1, 64);
        if (lane == 0) { rw = 0.0; rp = 0.0; }
#pragma unroll
        for (int k = 0; k < KW_PER; ++k) {
            const int e = lane * KW_PER + k;
            L.rank[e] = rk[k]; L.fw[e] = rw; L.fp[e] = rp;
            rw += wk[k]; rp += pk[k];
        }
        if (lane == 63) { L.rank[KW_CAP] = INT_MAX; L.fw[KW_CAP] = rw; L.fp[KW_CAP] = rp; }
    }
    __syncthreads();
    if (!live) return;
    const double W1n = wave_sum_f64(w1) + (x1val == 1 ? x1w : 0.0), P1n = wave_sum_f64(p1) + (x1val == 1 ? x1p : 0.0);
    // #entries of the parent's list with rank < t (per lane, its own t): binary search in LDS
    int s0 = 1; while (2 * s0 <= d) s0 *= 2;
    auto below = [&](int t) {
        int idx = 0;
        if (d > 0)
            for (int s = s0; s >= 1; s >>= 1)
                if (idx + s <= KW_CAP && L.rank[idx + s - 1] < t) idx += s;
        return idx;
    };
    const int before = below(x1pos);
    if (lane == 0) J.child[before] = (uint32_t)x1pos | ((uint32_t)x1val << 31);

    auto solve = [&](int xpos, int xval, double xw, double xp, int hint, int slot) {
        const double W1 = W1n + (xval == 1 ? xw : 0.0), P1 = P1n + (xval == 1 ? xp : 0.0);
        if (W1 > P.cap + KEPS) {                                        // :455-456
            if (lane == 0) { P.out_profit[slot] = P1; P.out_weight[slot] = W1; P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0; }
            return -1;
        }
        auto fixed_before = [&](int t, double& fw, double& fp) {
            const int c = below(t);
            fw = L.fw[c]; fp = L.fp[c];
            if (x1pos < t) { fw += x1w; fp += x1p; }
            if (xpos < t) { fw += xw; fp += xp; }
        };
        int lo = 1, hi = n + 1;                                         // t* in [lo, hi]; g(hi) holds or hi == n + 1
        // one round: lanes probe base + lane * stride (inside [lo, hi)), the ballots move lo / hi
        auto round = [&](int base, int stride) {
            const int t = base + lane * stride;
            const bool in = t >= lo && t < hi;
            bool over = false;
            if (in) {
                double fw, fp;
                fixed_before(t, fw, fp);
                over = W1 + (X.PW[t] - fw) > P.cap + KEPS;
            }
            const unsigned long long yes = __ballot(in && over), no = __ballot(in && !over);
            if (yes) hi = base + (int)__builtin_ctzll(yes) * stride;
            if (no) lo = base + (63 - (int)__builtin_clzll(no)) * stride + 1;
        };
        if (hint >= 1 && hint <= n) round(max(1, hint - 31), 1);
        while (lo < hi) {
            const int c = hi - lo, stride = (c + 63) >> 6;
            round(lo + stride - 1, stride);
        }
        double fw, fp;
        if (lo == n + 1) {
            fixed_before(n, fw, fp);
            if (lane == 0) {
                P.out_profit[slot] = P1 + (X.PP[n] - fp); P.out_weight[slot] = W1 + (X.PW[n] - fw);
                P.out_frac[slot] = -1; P.out_fracval[slot] = 0.0;
            }
            return -1;
        }
        const int j = lo - 1;
        fixed_before(j, fw, fp);
        double w = W1 + (X.PW[j] - fw), p = P1 + (X.PP[j] - fp);
        int frac = -1; double fv = 0.0;
        const double wi = P.ws[j];
        const double remain = P.cap - w;
        if (remain > KEPS && wi > KEPS) {                               // :476-484
            fv = remain / wi;
            p += P.ps[j] * fv;
            w += wi * fv;
            frac = j;
        }
        if (lane == 0) { P.out_profit[slot] = p; P.out_weight[slot] = w; P.out_frac[slot] = frac; P.out_fracval[slot] = fv; }
        return frac;
    };
    const int frac = solve(INT_MAX, 0, 0.0, 0.0, x1pos + 1, 3 * job);
    if (frac < 0) {
        if (lane == 0) { P.out_frac[3 * job + 1] = -2; P.out_frac[3 * job + 2] = -2; }
        return;
    }
    const double xw = P.ws[frac], xp = P.ps[frac];
    solve(frac, 0, xw, xp, frac + 1, 3 * job + 1);
    solve(frac, 1, xw, xp, frac + 1, 3 * job + 2);
    // the lists of the child's two children
    const int x2 = frac;
    const int before2 = below(x2);
#pragma unroll
    for (int k = 0; k < KW_PER; ++k) {
        const int e = lane * KW_PER + k;
        if (e < d) {
            const int dst = e + ((rk[k] > x1pos) ? 1 : 0) + ((rk[k] > x2) ? 1 : 0);
            J.g0[dst] = ev[k]; J.g1[dst] = ev[k];
        }
    }
    if (lane == 0) {
        const int q1 = before + ((x1pos > x2) ? 1 : 0), q2 = before2 + ((x2 > x1pos) ? 1 : 0);
        const uint32_t e1 = (uint32_t)x1pos | ((uint32_t)x1val << 31);
        J.g0[q1] = e1; J.g1[q1] = e1;
        J.g0[q2] = (uint32_t)x2; J.g1[q2] = (uint32_t)x2 | 0x80000000u;
    }
}

}  // namespace lpx

using namespace lpx;

struct lpx_knapsack {
    double *PW = nullptr, *PP = nullptr; bool prefix_ok = false;
    int n = 0; double cap = 0;
    double *ws = nullptr, *ps = nullptr, *w0 = nullptr, *p0 = nullptr;
    int32_t* pos = nullptr;
    std::vector<int32_t> order;
    // staging (grown on demand)
    int cap_nodes = 0, cap_fix = 0;
    int32_t *d_off = nullptr, *d_fidx = nullptr, *d_frac = nullptr; int8_t* d_fval = nullptr;
    double *d_profit = nullptr, *d_weight = nullptr, *d_fracval = nullptr;
    // pinned staging so that a batch is one H2D + one D2H instead of seven pageable copies
    char* d_in = nullptr; char* h_in = nullptr; size_t in_cap = 0;
    char* d_out = nullptr; char* h_out = nullptr; size_t out_cap = 0;
    hipStream_t stream = nullptr;
    // device-resident node store (lpx_knapsack_expand_batch): bump-allocated lists in 64 MiB chunks
    std::vector<uint32_t*> chunks; size_t chunk_used = 0;
    std::vector<uint32_t*> node_list; std::vector<int32_t> node_depth;      // node id -> (list, depth)
    char* d_jobs = nullptr; char* h_jobs = nullptr; size_t jobs_cap = 0;
    bool wide = true;                                                        // LPX_KNAP_WIDE=0: always the one-probe-per-step kernel
    int pending_out = 0;                                                     // results of the batch in flight (expand_begin .. expand_finish)
};

static constexpr size_t KN_CHUNK_WORDS = (size_t)16 << 20;                   // 64 MiB of uint32

extern "C" {

void lpx_knapsack_destroy(lpx_knapsack* k)
{
    if (!k) return;
    if (k->stream) hipStreamSynchronize(k->stream);
    hipFree(k->ws); hipFree(k->ps); hipFree(k->w0); hipFree(k->p0); hipFree(k->pos); hipFree(k->PW); hipFree(k->PP);
    hipFree(k->d_off); hipFree(k->d_fidx); hipFree(k->d_frac); hipFree(k->d_fval);
    hipFree(k->d_profit); hipFree(k->d_weight); hipFree(k->d_fracval);
    hipFree(k->d_in); hipFree(k->d_out); hipFree(k->d_jobs);
    for (uint32_t* c : k->chunks) hipFree(c);
    if (k->h_jobs) hipHostFree(k->h_jobs);
    if (k->h_in) hipHostFree(k->h_in);
    if (k->h_out) hipHostFree(k->h_out);
    delete k;
}

int lpx_knapsack_create(const double* profit, const double* weight, int n, double cap, lpx_knapsack** out)
{
    if (!profit || !weight || n < 1 || !out) { set_error("lpx_knapsack_create: bad argument"); return LPX_EINVAL; }
    if (n > 1200000) { set_error("lpx_knapsack_create: n above the LDS bitmap capacity (1.2M items)"); return LPX_EINVAL; }
    int rc = ensure_device();
    if (rc) return rc;
    static std::once_flag once; static hipError_t ierr = hipSuccess;
    std::call_once(once, [] {
        ierr = hipFuncSetAttribute(reinterpret_cast<const void*>(knap_relax_batch),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
    if (ierr != hipSuccess) { set_error("knap_relax_batch attribute setup failed"); return LPX_EDEVICE; }
    lpx_knapsack* k = new lpx_knapsack();
    k->n = n; k->cap = cap;
    // ratio order, Models/BranchAndBoundKnapsack.cs:19,75-79 (stable: index breaks remaining ties)
    std::vector<double> ratio(n);
    for (int i = 0; i < n; ++i) ratio[i] = weight[i] > 0 ? profit[i] / weight[i] : INFINITY;
    k->order.resize(n);
    for (int i = 0; i < n; ++i) k->order[i] = i;
    std::stable_sort(k->order.begin(), k->order.end(), [&](int a, int b) {
        if (ratio[a] != ratio[b]) return ratio[a] > ratio[b];
        return profit[a] > profit[b];
    });
    std::vector<double> ws(n), ps(n); std::vector<int32_t> pos(n);
    for (int s = 0; s < n; ++s) { ws[s] = weight[k->order[s]]; ps[s] = profit[k->order[s]]; pos[k->order[s]] = s; }
    hipError_t e = hipSuccess;
    auto up = [&](void** d, const void* h, size_t bytes) {
        if (e != hipSuccess) return;
        e = malloc_retry(d, bytes);
        if (e == hipSuccess) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    up((void**)&k->ws, ws.data(), sizeof(double) * n); up((void**)&k->ps, ps.data(), sizeof(double) * n);
    up((void**)&k->w0, weight, sizeof(double) * n); up((void**)&k->p0, profit, sizeof(double) * n);
    up((void**)&k->pos, pos.data(), sizeof(int32_t) * n);
    { const char* ev = std::getenv("LPX_KNAP_WIDE"); k->wide = !(ev && ev[0] == '0'); }
    // running sums in ratio order (left-to-right, as the reference accumulates) for the prefix-sum kernel
    std::vector<double> PW(n + 1, 0.0), PP(n + 1, 0.0);
    bool nonneg = true;
    for (int s = 0; s < n; ++s) { PW[s + 1] = PW[s] + ws[s]; PP[s + 1] = PP[s] + ps[s]; if (!(ws[s] >= 0.0)) nonneg = false; }
    k->prefix_ok = nonneg;
    up((void**)&k->PW, PW.data(), sizeof(double) * (n + 1)); up((void**)&k->PP, PP.data(), sizeof(double) * (n + 1));
    if (e == hipSuccess && (k->stream = borrow_stream()) == nullptr) e = hipErrorUnknown;
    if (e != hipSuccess) { set_error(std::string("lpx_knapsack_create: ") + hipGetErrorString(e)); lpx_knapsack_destroy(k); return LPX_EDEVICE; }
    *out = k;
    return 0;
}

int lpx_knapsack_order(lpx_knapsack* k, int32_t* order)
{
    if (!k || !order) return LPX_EINVAL;
    std::memcpy(order, k->order.data(), sizeof(int32_t) * k->n);
    return 0;
}

static int relax_batch_impl(lpx_knapsack* k, int count, const int32_t* off, const int32_t* fix_idx,
                            const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                            double* frac_val, bool depth2)
{
    if (!k || count < 0 || !off) { set_error("lpx_knapsack_relax_batch: bad argument"); return LPX_EINVAL; }
    if (count == 0) return 0;
    const int nout = depth2 ? 3 * count : count;
    const int nfix = off[count];
    for (int e = 0; e < nfix; ++e)
        if (fix_idx[e] < 0 || fix_idx[e] >= k->n) { set_error("lpx_knapsack_relax_batch: fixed index outside [0,n)"); return LPX_EINVAL; }
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_off = 0, o_fidx = up8(sizeof(int32_t) * (count + 1)), o_fval = o_fidx + up8(sizeof(int32_t) * (size_t)(nfix > 0 ? nfix : 1));
    const size_t in_bytes = o_fval + up8((size_t)(nfix > 0 ? nfix : 1));
    const size_t o_p = 0, o_w = sizeof(double) * nout, o_fv = 2 * sizeof(double) * nout, o_fr = 3 * sizeof(double) * nout;
    const size_t out_bytes = o_fr + up8(sizeof(int32_t) * nout);
    if (in_bytes > k->in_cap) {
        hipFree(k->d_in); if (k->h_in) hipHostFree(k->h_in);
        k->d_in = nullptr; k->h_in = nullptr; k->in_cap = 0;
        const size_t c = std::max(in_bytes, 2 * k->in_cap) + 4096;
        LPX_HIP_TRY(hipMalloc((void**)&k->d_in, c));
        LPX_HIP_TRY(hipHostMalloc((void**)&k->h_in, c));
        k->in_cap = c;
    }
    if (out_bytes > k->out_cap) {
        hipFree(k->d_out); if (k->h_out) hipHostFree(k->h_out);
        k->d_out = nullptr; k->h_out = nullptr; k->out_cap = 0;
        const size_t c = std::max(out_bytes, 2 * k->out_cap) + 4096;
        LPX_HIP_TRY(hipMalloc((void**)&k->d_out, c));
        LPX_HIP_TRY(hipHostMalloc((void**)&k->h_out, c));
        k->out_cap = c;
    }
    std::memcpy(k->h_in + o_off, off, sizeof(int32_t) * (count + 1));
    if (nfix > 0) { std::memcpy(k->h_in + o_fidx, fix_idx, sizeof(int32_t) * nfix); std::memcpy(k->h_in + o_fval, fix_val, nfix); }
    hipStream_t s = k->stream;
    LPX_HIP_TRY(hipMemcpyAsync(k->d_in, k->h_in, in_bytes, hipMemcpyHostToDevice, s));
    KnParams P;
    P.n = k->n; P.cap = k->cap; P.ws = k->ws; P.ps = k->ps; P.pos = k->pos; P.w0 = k->w0; P.p0 = k->p0;
    P.count = count;
    P.off = reinterpret_cast<const int32_t*>(k->d_in + o_off);
    P.fidx = reinterpret_cast<const int32_t*>(k->d_in + o_fidx);
    P.fval = reinterpret_cast<const int8_t*>(k->d_in + o_fval);
    P.out_profit = reinterpret_cast<double*>(k->d_out + o_p); P.out_weight = reinterpret_cast<double*>(k->d_out + o_w);
    P.out_fracval = reinterpret_cast<double*>(k->d_out + o_fv); P.out_frac = reinterpret_cast<int32_t*>(k->d_out + o_fr);
    static const bool force_scan = [] { const char* e = std::getenv("LPX_KNAP_SCAN"); return e && e[0] == '1'; }();
    if (k->prefix_ok && !force_scan) {
        KnPrefix X; X.PW = k->PW; X.PP = k->PP;
        if (depth2) hipLaunchKernelGGL(knap_relax_prefix<true>, dim3((count + 3) / 4), dim3(256), 0, s, P, X);
        else hipLaunchKernelGGL(knap_relax_prefix<false>, dim3((count + 3) / 4), dim3(256), 0, s, P, X);
    } else if (depth2) {
        set_error("lpx_knapsack_relax_batch2 needs the prefix-sum path (non-negative weights)"); return LPX_EINVAL;
    } else {
        const size_t dyn = sizeof(unsigned int) * ((k->n + 31) / 32);
        hipLaunchKernelGGL(knap_relax_batch, dim3(count), dim3(KN_NT), dyn, s, P);
    }
    LPX_HIP_TRY(hipGetLastError());
    LPX_HIP_TRY(hipMemcpyAsync(k->h_out, k->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    LPX_HIP_TRY(hipStreamSynchronize(s));
    if (profit) std::memcpy(profit, k->h_out + o_p, sizeof(double) * nout);
    if (weight) std::memcpy(weight, k->h_out + o_w, sizeof(double) * nout);
    if (frac_val) std::memcpy(frac_val, k->h_out + o_fv, sizeof(double) * nout);
    if (frac_idx) std::memcpy(frac_idx, k->h_out + o_fr, sizeof(int32_t) * nout);
    return 0;
}

int lpx_knapsack_relax_batch(lpx_knapsack* k, int count, const int32_t* off, const int32_t* fix_idx,
                             const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                             double* frac_val)
{
    return relax_batch_impl(k, count, off, fix_idx, fix_val, profit, weight, frac_idx, frac_val, false);
}

int lpx_knapsack_relax_batch2(lpx_knapsack* k, int count, const int32_t* off, const int32_t* fix_idx,
                              const int8_t* fix_val, double* profit, double* weight, int32_t* frac_idx,
                              double* frac_val)
{
    if (k && !k->prefix_ok) { set_error("lpx_knapsack_relax_batch2 needs the prefix-sum path (non-negative weights)"); return LPX_EINVAL; }
    return relax_batch_impl(k, count, off, fix_idx, fix_val, profit, weight, frac_idx, frac_val, true);
}

int lpx_knapsack_has_prefix(lpx_knapsack* k) { return k && k->prefix_ok ? 1 : 0; }

int lpx_knapsack_expand_begin(lpx_knapsack* k, int count, const int64_t* parent, const int32_t* item, const int8_t* val, int64_t* child)
{
    if (!k || count < 0 || (count > 0 && (!parent || !item || !val || !child))) { set_error("lpx_knapsack_expand_begin: bad argument"); return LPX_EINVAL; }
    if (!k->prefix_ok) { set_error("lpx_knapsack_expand_batch needs the prefix-sum path (non-negative weights)"); return LPX_EINVAL; }
    if (k->pending_out) { set_error("lpx_knapsack_expand_begin: a batch is already in flight (lpx_knapsack_expand_finish first)"); return LPX_EINVAL; }
    if (count == 0) return 0;
    const int64_t known = (int64_t)k->node_list.size();
    for (int j = 0; j < count; ++j) {
        if (parent[j] < -1 || parent[j] >= known || item[j] < 0 || item[j] >= k->n || (val[j] != 0 && val[j] != 1)) {
            set_error("lpx_knapsack_expand_batch: parent id, item or value out of range"); return LPX_EINVAL; }
    }
    const size_t jb = sizeof(KnJob) * (size_t)count;
    if (jb > k->jobs_cap) {
        hipFree(k->d_jobs); if (k->h_jobs) hipHostFree(k->h_jobs);
        k->d_jobs = nullptr; k->h_jobs = nullptr; k->jobs_cap = 0;
        const size_t c = 2 * jb + 4096;
        LPX_HIP_TRY(hipMalloc((void**)&k->d_jobs, c));
        LPX_HIP_TRY(hipHostMalloc((void**)&k->h_jobs, c));
        k->jobs_cap = c;
    }
    KnJob* hj = reinterpret_cast<KnJob*>(k->h_jobs);
    auto carve = [&](size_t words, uint32_t** out) -> int {
        const size_t need = (words + 3) & ~(size_t)3;                        // 16-byte granules
        if (need > KN_CHUNK_WORDS) { set_error("lpx_knapsack_expand_batch: node deeper than a store chunk"); return LPX_EINVAL; }
        if (k->chunks.empty() || k->chunk_used + need > KN_CHUNK_WORDS) {
            uint32_t* c = nullptr;
            if (malloc_retry((void**)&c, sizeof(uint32_t) * KN_CHUNK_WORDS) != hipSuccess) { set_error("lpx_knapsack_expand_batch: out of device memory for the node store"); return LPX_ENOMEM; }
            k->chunks.push_back(c); k->chunk_used = 0;
        }
        *out = k->chunks.back() + k->chunk_used;
        k->chunk_used += need;
        return 0;
    };
    int maxdepth = 0;
    for (int j = 0; j < count; ++j) {
        const int d = parent[j] < 0 ? 0 : k->node_depth[(size_t)parent[j]];
        maxdepth = std::max(maxdepth, d);
        uint32_t *c0 = nullptr, *g0 = nullptr, *g1 = nullptr;
        int rc = carve((size_t)d + 1, &c0); if (rc) return rc;
        rc = carve((size_t)d + 2, &g0); if (rc) return rc;
        rc = carve((size_t)d + 2, &g1); if (rc) return rc;
        hj[j].parent = parent[j] < 0 ? c0 : k->node_list[(size_t)parent[j]];    // depth 0: never dereferenced
        hj[j].child = c0; hj[j].g0 = g0; hj[j].g1 = g1; hj[j].depth = d; hj[j].item = item[j]; hj[j].val = val[j]; hj[j].pad = 0;
        child[j] = (int64_t)k->node_list.size();                             // ids child[j] + 1 / + 2: its two children (if any)
        k->node_list.push_back(c0); k->node_depth.push_back(d + 1);
        k->node_list.push_back(g0); k->node_depth.push_back(d + 2);
        k->node_list.push_back(g1); k->node_depth.push_back(d + 2);
    }
    const int nout = 3 * count;
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_p = 0, o_w = sizeof(double) * nout, o_fv = 2 * sizeof(double) * nout, o_fr = 3 * sizeof(double) * nout;
    const size_t out_bytes = o_fr + up8(sizeof(int32_t) * nout);
    if (out_bytes > k->out_cap) {
        hipFree(k->d_out); if (k->h_out) hipHostFree(k->h_out);
        k->d_out = nullptr; k->h_out = nullptr; k->out_cap = 0;
        const size_t c = 2 * out_bytes + 4096;
        LPX_HIP_TRY(hipMalloc((void**)&k->d_out, c));
        LPX_HIP_TRY(hipHostMalloc((void**)&k->h_out, c));
        k->out_cap = c;
    }
    // Zero-copy: the kernel reads its 48-byte job records from, and writes its results to, pinned host memory directly
    // (hipHostMalloc memory is device-visible and coherent).  A batch is then ONE launch and one wait instead of
    // copy + launch + copy + wait: the loop is bound by that round trip, not by bytes (a few KB per batch).
    hipStream_t s = k->stream;
    KnParams P;
    P.n = k->n; P.cap = k->cap; P.ws = k->ws; P.ps = k->ps; P.pos = k->pos; P.w0 = k->w0; P.p0 = k->p0;
    P.count = count; P.off = nullptr; P.fidx = nullptr; P.fval = nullptr;
    P.out_profit = reinterpret_cast<double*>(k->h_out + o_p); P.out_weight = reinterpret_cast<double*>(k->h_out + o_w);
    P.out_fracval = reinterpret_cast<double*>(k->h_out + o_fv); P.out_frac = reinterpret_cast<int32_t*>(k->h_out + o_fr);
    KnPrefix X; X.PW = k->PW; X.PP = k->PP;
    // every node of the batch within the wide kernel's list capacity -> 64 probes per round; deeper nodes -> one probe per step
    if (maxdepth <= KW_CAP && k->wide)
        hipLaunchKernelGGL(knap_expand_w, dim3((count + 3) / 4), dim3(256), 0, s, P, X, reinterpret_cast<const KnJob*>(k->h_jobs));
    else
        hipLaunchKernelGGL(knap_expand, dim3((count + 3) / 4), dim3(256), 0, s, P, X, reinterpret_cast<const KnJob*>(k->h_jobs));
    LPX_HIP_TRY(hipGetLastError());
    k->pending_out = nout;
    return 0;
}

// Waits for the batch lpx_knapsack_expand_begin enqueued and hands out its 3 * count results (slot layout as
// lpx_knapsack_expand_batch).  The host is free between the two calls: the jobs and the results live in pinned memory the
// kernel reads and writes directly.
int lpx_knapsack_expand_finish(lpx_knapsack* k, double* profit, double* weight, int32_t* frac_idx, double* frac_val)
{
    if (!k) { set_error("lpx_knapsack_expand_finish: null handle"); return LPX_EINVAL; }
    const int nout = k->pending_out;
    if (nout <= 0) return 0;
    LPX_HIP_TRY(hipStreamSynchronize(k->stream));
    k->pending_out = 0;
    const size_t o_p = 0, o_w = sizeof(double) * nout, o_fv = 2 * sizeof(double) * nout, o_fr = 3 * sizeof(double) * nout;
    if (profit) std::memcpy(profit, k->h_out + o_p, sizeof(double) * nout);
    if (weight) std::memcpy(weight, k->h_out + o_w, sizeof(double) * nout);
    if (frac_val) std::memcpy(frac_val, k->h_out + o_fv, sizeof(double) * nout);
    if (frac_idx) std::memcpy(frac_idx, k->h_out + o_fr, sizeof(int32_t) * nout);
    return 0;
}

int lpx_knapsack_expand_batch(lpx_knapsack* k, int count, const int64_t* parent, const int32_t* item, const int8_t* val,
                              int64_t* child, double* profit, double* weight, int32_t* frac_idx, double* frac_val)
{
    if (k && k->pending_out) { set_error("lpx_knapsack_expand_batch: a batch is in flight (lpx_knapsack_expand_finish first)"); return LPX_EINVAL; }
    int rc = lpx_knapsack_expand_begin(k, count, parent, item, val, child);
    if (rc) return rc;
    return lpx_knapsack_expand_finish(k, profit, weight, frac_idx, frac_val);
}

int lpx_knapsack_node_list(lpx_knapsack* k, int64_t node, int32_t* idx, int8_t* val, int cap, int* depth)
{
    if (!k || node < 0 || node >= (int64_t)k->node_list.size()) { set_error("lpx_knapsack_node_list: unknown node id"); return LPX_EINVAL; }
    const int d = k->node_depth[(size_t)node];
    if (depth) *depth = d;
    if (idx && val && cap > 0 && d > 0) {
        std::vector<uint32_t> tmp((size_t)d);
        LPX_HIP_TRY(hipStreamSynchronize(k->stream));
        LPX_HIP_TRY(hipMemcpy(tmp.data(), k->node_list[(size_t)node], sizeof(uint32_t) * d, hipMemcpyDeviceToHost));
        // the store keeps ratio ranks; the caller gets item indices in ascending order (the reference's Assigned order)
        std::vector<std::pair<int32_t, int8_t>> ent((size_t)d);
        for (int e = 0; e < d; ++e) ent[(size_t)e] = {k->order[(size_t)(tmp[e] & 0x7fffffffu)], (int8_t)(tmp[e] >> 31)};
        std::sort(ent.begin(), ent.end());
        for (int e = 0; e < d && e < cap; ++e) { idx[e] = ent[(size_t)e].first; val[e] = ent[(size_t)e].second; }
    }
    return 0;
}

}  // extern "C"
