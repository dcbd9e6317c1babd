// host/knapsack.cpp -- BranchAndBoundKnapsack mirror (Models/BranchAndBoundKnapsack.cs:58-407).
//
// The search is the reference's: best-first on the greedy fractional bound with its own array heap
// (Push sift-up breaks on `<= 0`, Pop swaps the last element in and sifts down, :494-547), left child
// (x=0) evaluated before right child (x=1), incumbent rule `> best + 1e-9`.  What moves to the GPU is
// ComputeRelaxation -- in bulk and AHEAD of the search:
//   * every relaxation the device has produced is a node of an "evaluated tree" (KNode: parent + one decision + its
//     bound); the reference's heap holds pointers into that tree, so pushing a child allocates and computes nothing;
//   * the evaluated leaves (bound known, children not) sit in a second max-heap keyed by the same bound the search pops
//     by.  Best-first pops in bound order, so the leaves with the largest bounds ARE the nodes whose children the search
//     asks for next: whenever the loop reaches a node whose children are missing, one launch evaluates that node and
//     the `spec` best leaves (each job: a child and that child's two children, lpx_knapsack_expand_batch);
//   * a relaxation depends only on the node's fixed set, never on the search order, so evaluating early cannot change
//     any decision: the host replays the exact reference order (popped / expanded / relaxations counts equal the CPU
//     restatement's) while the device sees batches of hundreds of jobs instead of one node at a time.
// Report text (3n lines per expansion, :139-143) is not produced.
//
// Sharded form (world > 1): the tree is expanded redundantly until the heap holds >= 4*world nodes,
// heap[i] then belongs to rank i % world; every rank runs the same best-first loop on its own heap for
// `round` pops, then ONE all-reduce(max) over {incumbent, have_work} (X1) and the loop continues.
#include "model.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>

namespace lpx { namespace host {

namespace {

constexpr double EPS = 1e-9;          // :56

std::string last_error() { char b[1024]; lpx_last_error(b, sizeof(b)); return b; }

struct Relax { double profit = 0, weight = 0, fracval = 0; int frac = -1; bool valid = false; };

struct KNode {
    // a node is its parent plus one decision (:207-209, :267-269); the fixed list itself lives in the device store
    // (lpx_knapsack_expand_batch) under `dev`, or is rebuilt from this chain when a host-side list is needed
    KNode* parent = nullptr;
    KNode* kid[2] = {nullptr, nullptr};                    // evaluated children (x=0, x=1); null = not evaluated yet
    int32_t item = -1; int32_t depth = 0;
    int64_t dev = -1;                                      // id in the device store (-1 = the root / no store)
    Relax self; double bound = 0;                          // own relaxation (bound == self.profit)
    int8_t val = 0; bool queued = false;                   // queued: its children are part of the launch being assembled
    // fixed decisions in ascending item index (the reference's Assigned without the undecided entries)
    void list(std::vector<int32_t>& idx, std::vector<int8_t>& val) const {
        std::vector<std::pair<int32_t, int8_t>> e; e.reserve((size_t)depth);
        for (const KNode* p = this; p && p->item >= 0; p = p->parent) e.emplace_back(p->item, p->val);
        std::sort(e.begin(), e.end());
        idx.clear(); val.clear();
        for (auto& x : e) { idx.push_back(x.first); val.push_back(x.second); }
    }
};

struct Heap {                                              // SimpleMaxHeap<Node>, :494-547
    // the key sits beside the pointer: a sift compares array entries only (the reference compares node.Bound through the
    // reference, same values) -- with 10^5..10^6 nodes the pointer chase was most of the host time per pop
    struct E { double bound; KNode* n; };
    std::vector<E> d;
    static int cmp(const E& a, const E& b) { return (a.bound > b.bound) - (a.bound < b.bound); }
    void push(KNode* x) {
        d.push_back(E{x->bound, x});
        size_t ci = d.size() - 1;
        while (ci > 0) { size_t pi = (ci - 1) / 2; if (cmp(d[ci], d[pi]) <= 0) break; std::swap(d[ci], d[pi]); ci = pi; }
    }
    KNode* pop() {
        size_t li = d.size() - 1;
        std::swap(d[0], d[li]);
        KNode* ret = d[li].n; d.pop_back();
        if (d.empty()) return ret;
        li = d.size() - 1;
        size_t i = 0;
        for (;;) {
            size_t l = 2 * i + 1, r = 2 * i + 2, largest = i;
            // the four grandchildren are 64 contiguous bytes: have them on their way while this level is compared
            if (4 * i + 3 <= li) __builtin_prefetch(&d[4 * i + 3]);
            if (l <= li && cmp(d[l], d[largest]) > 0) largest = l;
            if (r <= li && cmp(d[r], d[largest]) > 0) largest = r;
            if (largest == i) break;
            std::swap(d[i], d[largest]); i = largest;
        }
        return ret;
    }
};

struct Search {
    lpx_knapsack* k; int n; double cap;
    std::vector<double> profit, weight; std::vector<int32_t> order;
    double best = -INFINITY; std::vector<int32_t> bestX; bool has_best = false;
    int64_t popped = 0, expanded = 0, relaxations = 0, max_heap = 0, launches = 0, jobs_run = 0;
    int64_t redundant_popped = 0, redundant_relax = 0;   // replicated warm-up on ranks != 0
    double dev_ms = 0;                                   // wall time inside the lpx_knapsack_* calls (launch + wait)
    int spec = 256;                                      // evaluated leaves expanded ahead of the search per launch
    std::function<int(int, const int32_t*, const int32_t*, const int8_t*, double*, double*, int32_t*, double*)> test_relax;
    // the evaluated tree: nodes are carved from raw 64k-node slabs and every field is written by the caller (root: eval_root,
    // others: integrate) -- three nodes per job make a constructor call and a deque bookkeeping step per node measurable
    std::vector<KNode*> slabs; size_t slab_used = 0;
    static constexpr size_t SLAB = (size_t)1 << 16;
    ~Search() { for (KNode* p : slabs) std::free(p); }
    Heap leaves;                                         // evaluated, unexpanded, still worth expanding: keyed by bound
    bool depth2 = false;                                 // one launch also evaluates each job's two children
    bool use_store = false;                              // device-resident node lists (lpx_knapsack_expand_batch)

    KNode* alloc()
    {
        if (slabs.empty() || slab_used == SLAB) {
            KNode* p = static_cast<KNode*>(std::malloc(sizeof(KNode) * SLAB));
            if (!p) throw LpxException(LPX_ENOMEM, "knapsack: out of host memory for the evaluated tree");
            slabs.push_back(p); slab_used = 0;
        }
        KNode* x = slabs.back() + slab_used++;
        x->kid[0] = nullptr; x->kid[1] = nullptr; x->queued = false;
        return x;
    }
    bool worth_expanding(const KNode* x) const           // would the search branch on it if it popped it now?
    { return x->self.frac >= 0 && x->self.weight <= cap + EPS && x->self.profit > best + EPS; }
    void offer_leaf(KNode* x) { if (!x->kid[0] && !x->kid[1] && worth_expanding(x)) leaves.push(x); }

    struct Clock { double& acc; std::chrono::steady_clock::time_point t0;
                   explicit Clock(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
                   ~Clock() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } };

    void eval_root(KNode* root)
    {
        double p = 0, w = 0, fv = 0; int32_t fr = -1;
        const int32_t off[2] = {0, 0}; const int32_t fidx[1] = {0}; const int8_t fval[1] = {0};
        int rc;
        { Clock c(dev_ms);
          rc = test_relax ? test_relax(1, off, fidx, fval, &p, &w, &fr, &fv) : lpx_knapsack_relax_batch(k, 1, off, fidx, fval, &p, &w, &fr, &fv); }
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        ++launches;
        root->self.profit = p; root->self.weight = w; root->self.frac = fr; root->self.fracval = fv; root->self.valid = true;
        root->bound = p;
    }

    // one launch: child `v` of every job's node (its fractional item fixed to v) and, with depth2, that child's two children
    struct Job { KNode* node; int v; };
    // a batch in flight on the device (store path): its jobs and the ids the device gave the new nodes
    std::vector<Job> flying; std::vector<int64_t> flying_ids;
    bool in_flight() const { return !flying.empty(); }
    int64_t n_async = 0, n_sync = 0, n_async_hit = 0; double wait_ms = 0, begin_ms = 0;   // diagnostics (LPX_KNAP_DEBUG=1)

    void integrate(const std::vector<Job>& jobs, const std::vector<int64_t>& ch, const std::vector<double>& p, const std::vector<double>& w,
                   const std::vector<int32_t>& fr, const std::vector<double>& fv)
    {
        const size_t st = depth2 ? 3 : 1;
        ++launches; jobs_run += (int64_t)jobs.size();
        auto fill = [&](KNode* x, KNode* parent, int item, int v, size_t o, int64_t dev) {
            x->parent = parent; x->item = item; x->val = (int8_t)v; x->depth = parent->depth + 1; x->dev = dev;
            x->self.profit = p[o]; x->self.weight = w[o]; x->self.frac = fr[o]; x->self.fracval = fv[o]; x->self.valid = true;
            x->bound = p[o];
        };
        for (size_t j = 0; j < jobs.size(); ++j) {
            KNode* nd = jobs[j].node; const int v = jobs[j].v;
            KNode* c = alloc();
            fill(c, nd, order[nd->self.frac], v, st * j, ch[j]);
            nd->kid[v] = c; nd->queued = false;
            if (depth2 && c->self.frac >= 0) {
                for (int cc = 0; cc < 2; ++cc) {
                    const size_t o = st * j + 1 + (size_t)cc;
                    if (fr[o] == -2) continue;                  // the child has nothing to branch on
                    KNode* g = alloc();
                    fill(g, c, order[c->self.frac], cc, o, ch[j] < 0 ? -1 : ch[j] + 1 + cc);
                    c->kid[cc] = g;
                    if (worth_expanding(c)) offer_leaf(g);
                }
            } else if (!depth2) offer_leaf(c);
        }
    }

    // store path, asynchronous half: enqueue the batch and return -- the device works while the host replays the search
    void begin_jobs(std::vector<Job>& jobs)
    {
        std::vector<int64_t> par(jobs.size()); std::vector<int32_t> it(jobs.size()); std::vector<int8_t> vv(jobs.size());
        for (size_t j = 0; j < jobs.size(); ++j) { par[j] = jobs[j].node->dev; it[j] = order[jobs[j].node->self.frac]; vv[j] = (int8_t)jobs[j].v; }
        flying_ids.assign(jobs.size(), -1);
        int rc;
        { Clock c(dev_ms); Clock c2(begin_ms); rc = lpx_knapsack_expand_begin(k, (int)jobs.size(), par.data(), it.data(), vv.data(), flying_ids.data()); }
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        flying.swap(jobs);
    }
    void finish_jobs()
    {
        if (flying.empty()) return;
        const size_t nout = 3 * flying.size();
        std::vector<double> p(nout), w(nout), fv(nout); std::vector<int32_t> fr(nout);
        int rc;
        { Clock c(dev_ms); Clock c2(wait_ms); rc = lpx_knapsack_expand_finish(k, p.data(), w.data(), fr.data(), fv.data()); }
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        std::vector<Job> jobs; jobs.swap(flying);
        integrate(jobs, flying_ids, p, w, fr, fv);
    }

    void run_jobs(std::vector<Job>& jobs)
    {
        if (jobs.empty()) return;
        if (use_store) { begin_jobs(jobs); finish_jobs(); return; }
        const size_t st = depth2 ? 3 : 1, nout = st * jobs.size();
        std::vector<double> p(nout), w(nout), fv(nout); std::vector<int32_t> fr(nout);
        std::vector<int64_t> ch(jobs.size(), -1);
        std::vector<int32_t> off(jobs.size() + 1, 0), fidx, li; std::vector<int8_t> fval, lv;
        for (size_t j = 0; j < jobs.size(); ++j) {
            jobs[j].node->list(li, lv);                  // ascending index order (:442)
            const int it = order[jobs[j].node->self.frac];
            bool placed = false;
            for (size_t e = 0; e < li.size(); ++e) {
                if (!placed && it < li[e]) { fidx.push_back(it); fval.push_back((int8_t)jobs[j].v); placed = true; }
                fidx.push_back(li[e]); fval.push_back(lv[e]);
            }
            if (!placed) { fidx.push_back(it); fval.push_back((int8_t)jobs[j].v); }
            off[j + 1] = (int32_t)fidx.size();
        }
        int rc;
        { Clock c(dev_ms);
          rc = test_relax ? test_relax((int)jobs.size(), off.data(), fidx.data(), fval.data(), p.data(), w.data(), fr.data(), fv.data())
             : depth2 ? lpx_knapsack_relax_batch2(k, (int)jobs.size(), off.data(), fidx.data(), fval.data(), p.data(), w.data(), fr.data(), fv.data())
                      : lpx_knapsack_relax_batch(k, (int)jobs.size(), off.data(), fidx.data(), fval.data(), p.data(), w.data(), fr.data(), fv.data()); }
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        integrate(jobs, ch, p, w, fr, fv);
    }

    // the best evaluated leaves, two jobs each
    int take_leaves(std::vector<Job>& jobs, int want)
    {
        int taken = 0;
        while (taken < want && !leaves.d.empty()) {
            KNode* o = leaves.pop();
            if (o->queued || o->kid[0] || o->kid[1] || !worth_expanding(o)) continue;   // stale entry
            o->queued = true;
            jobs.push_back({o, 0}); jobs.push_back({o, 1});
            ++taken;
        }
        return taken;
    }

    // relaxed vector of an evaluated node (host, O(n); only on incumbent updates)
    std::vector<double> relaxed_of(const KNode& nd) const
    {
        const Relax& r = nd.self;
        std::vector<int8_t> as(n, -1);
        for (const KNode* q = &nd; q && q->item >= 0; q = q->parent) as[q->item] = q->val;
        std::vector<double> x(n, 0.0);
        for (int i = 0; i < n; ++i) if (as[i] == 1) x[i] = 1.0;
        double fixedw = 0; for (int i = 0; i < n; ++i) if (as[i] == 1) fixedw += weight[i];
        if (fixedw > cap + EPS) return x;                   // :455-456
        const int stop = r.frac >= 0 ? r.frac : n;
        // items before the break in ratio order that are undecided were taken whole; when there is no
        // fractional item the greedy may still have stopped early (remain <= EPS): replay it
        double w = fixedw;
        for (int s = 0; s < n; ++s) {
            const int o = order[s];
            if (as[o] != -1) continue;
            if (s == stop) { x[o] = r.fracval; break; }
            if (w + weight[o] <= cap + EPS) { x[o] = 1.0; w += weight[o]; } else break;
        }
        return x;
    }

    void consider_child(Heap& pq, KNode* node, int v)
    {
        KNode* ch = node->kid[v];
        const Relax& r = ch->self;
        if (r.weight > cap + EPS) return;                                   // :215 / :277 INFEASIBLE
        if (r.profit > best + EPS) {                                        // :223 / :285
            const bool allInt = (r.frac < 0) || std::fabs(r.fracval - std::nearbyint(r.fracval)) < EPS;
            const bool feasible = r.weight <= cap + EPS;
            if (allInt && feasible) {                                       // :228-236
                if (r.profit > best + EPS) {
                    best = r.profit; has_best = true;
                    std::vector<double> x = relaxed_of(*ch);
                    bestX.assign(n, 0);
                    for (int i = 0; i < n; ++i) bestX[i] = (int32_t)std::nearbyint(x[i]);
                }
            } else {                                                        // :239-248
                pq.push(ch);                                                // already evaluated, its children possibly too
                max_heap = std::max<int64_t>(max_heap, (int64_t)pq.d.size());
            }
        }
        // else: logged as "CANDIDATE" and dropped (:257-264)
    }

    // one pop of the main loop (:118-328); returns false when the heap is empty
    bool step(Heap& pq)
    {
        if (pq.d.empty()) return false;
        KNode* node = pq.pop();
        ++popped;                                                           // :121
        if (node->bound <= best + EPS) return true;                         // :124
        ++expanded;
        ++relaxations;                                                      // :127 recompute (identical to the stored one)
        const Relax& self = node->self;
        if (self.frac == -1) {                                              // :147-177
            if (self.weight <= cap + EPS && self.profit > best + EPS) {
                best = self.profit; has_best = true;
                std::vector<double> x = relaxed_of(*node);
                bestX.assign(n, 0);
                for (int i = 0; i < n; ++i) bestX[i] = x[i] >= 0.5 ? 1 : 0;
            }
            return true;
        }
        if ((!node->kid[0] || !node->kid[1]) && in_flight()) finish_jobs();      // it may be in the batch that is on the device
        if (!node->kid[0] || !node->kid[1]) {
            // one launch: this node's children plus those of the best evaluated leaves -- the nodes the search pops next
            std::vector<Job> jobs;
            for (int v = 0; v < 2; ++v) if (!node->kid[v]) jobs.push_back({node, v});
            node->queued = true;
            take_leaves(jobs, spec);
            run_jobs(jobs); ++n_sync;
        }
        // keep the device busy while the host replays: as soon as enough leaves have gathered, their expansion is enqueued and
        // collected only when the search reaches one of them (or the next batch is due)
        if (use_store && !in_flight() && (int)leaves.d.size() >= spec) {
            std::vector<Job> jobs;
            if (take_leaves(jobs, spec) > 0) { begin_jobs(jobs); ++n_async; }
        }
        relaxations += 2;
        consider_child(pq, node, 0);                                        // LEFT  x=0, :207-264
        consider_child(pq, node, 1);                                        // RIGHT x=1, :267-327
        return true;
    }

    // after the frontier is split over the ranks: speculate only below nodes this rank owns
    void reseed_leaves(const Heap& pq)
    {
        leaves.d.clear();
        std::function<void(KNode*)> walk = [&](KNode* x) {
            if (!x->kid[0] && !x->kid[1]) { offer_leaf(x); return; }
            for (int v = 0; v < 2; ++v) if (x->kid[v] && worth_expanding(x)) walk(x->kid[v]);
        };
        for (auto& e : pq.d) walk(e.n);
    }
};

}  // namespace

SimplexResult BranchAndBoundKnapsack::Solve(const LPProblem& problem, UpdatePivot updatePivot)
{
    if (problem.Constraints.size() != 1)                                    // :66-67
        throw LpxException(LPX_E_KNAP_SHAPE, "Knapsack solver requires exactly one constraint (weights and capacity).");
    const Constraint& cons = problem.Constraints[0];
    if (cons.Relation != Rel::LE) throw LpxException(LPX_E_KNAP_SHAPE, "Knapsack solver requires a <= constraint.");   // :69
    const int n = problem.NumVars();
    if ((int)cons.A.size() < n) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");
    if (n < 1) throw LpxException(LPX_EINVAL, "Knapsack solver needs at least one item.");

    Search S;
    S.n = n; S.cap = cons.B; S.profit = problem.C; S.weight.assign(cons.A.begin(), cons.A.begin() + n);
    lpx_knapsack* kh = nullptr;
    S.test_relax = opt.test_knap_relax;
    S.order.resize(n);
    if (!S.test_relax) {
        int rc = lpx_knapsack_create(S.profit.data(), S.weight.data(), n, S.cap, &kh);
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        lpx_knapsack_order(kh, S.order.data());
    } else {
        // test seam: same ratio order as lpx_knapsack_create (Models/BranchAndBoundKnapsack.cs:19,75-79)
        std::vector<double> ratio(n);
        for (int i = 0; i < n; ++i) ratio[i] = S.weight[i] > 0 ? S.profit[i] / S.weight[i] : INFINITY;
        for (int i = 0; i < n; ++i) S.order[i] = i;
        std::stable_sort(S.order.begin(), S.order.end(), [&](int a, int b) {
            if (ratio[a] != ratio[b]) return ratio[a] > ratio[b];
            return S.profit[a] > S.profit[b];
        });
    }
    struct Guard { lpx_knapsack* k; ~Guard() { lpx_knapsack_destroy(k); } } guard{kh};
    S.k = kh;
    S.bestX.assign(n, 0);
    S.spec = opt.concurrent_nodes > 1 ? opt.concurrent_nodes : 256;
    {   // every launch also evaluates the children of the nodes it evaluates (LPX_KNAP_DEPTH2=0: off)
        const char* e = std::getenv("LPX_KNAP_DEPTH2");
        S.depth2 = !S.test_relax && kh && lpx_knapsack_has_prefix(kh) && !(e && e[0] == '0');
        // node lists resident on the device (LPX_KNAP_STORE=0: every job ships its whole list, the r01 path)
        const char* e2 = std::getenv("LPX_KNAP_STORE");
        S.use_store = S.depth2 && !(e2 && e2[0] == '0');
    }

    Heap pq;
    KNode* root = S.alloc();                                                // :102-113
    root->parent = nullptr; root->item = -1; root->val = 0; root->depth = 0; root->dev = -1;
    S.eval_root(root); S.relaxations++;
    pq.push(root);
    S.max_heap = 1;

    const int world = std::max(1, opt.world), rank = opt.rank;
    const int64_t cap_nodes = opt.max_nodes;
    bool replicated = world > 1;
    const bool sharded = (world > 1 || opt.shard_one) && (bool)opt.allreduce_max;
    const int round = 256;
    bool failed = false; std::string fail_msg; int fail_code = 0;
    // fingerprint of the heap as the replicated warm-up leaves it: {+h, -h} in every rank's FIRST all-reduce, max(+h) == -max(-h) iff
    // all ranks agree -- ranks that do not (a device path whose result depends on scheduling, as in round 2's record scale7) return an
    // error instead of waiting in collectives that no longer match (the same device as csrc/host/bnb.cpp frontier_fingerprint)
    auto fingerprint = [&](bool handed_out) {
        uint64_t h = 1469598103934665603ull ^ (handed_out ? 0x9e3779b97f4a7c15ull : 0ull);
        auto mix = [&](uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); };
        mix((uint64_t)pq.d.size()); mix((uint64_t)S.popped);
        for (const auto& e : pq.d) { int64_t b; double d = e.n->bound; std::memcpy(&b, &d, sizeof(b)); mix((uint64_t)b); mix((uint64_t)(uint32_t)e.n->item); mix((uint64_t)(uint32_t)e.n->depth); }
        return (double)(h & ((1ull << 52) - 1));
    };
    double agree_h = -1.0; bool agree_sent = !sharded;
    if (sharded && !replicated) agree_h = fingerprint(false);
    for (;;) {
        if (replicated && pq.d.size() >= (size_t)(4 * world)) {
            agree_h = fingerprint(true);
            Heap mine;
            for (size_t i = 0; i < pq.d.size(); ++i) if ((int)(i % world) == rank) mine.push(pq.d[i].n);
            pq.d.swap(mine.d);
            replicated = false;
            S.finish_jobs();
            S.reseed_leaves(pq);
            if (rank != 0) { S.redundant_popped = S.popped; S.redundant_relax = S.relaxations; }   // rank 0 accounts for the warm-up
        }
        bool more = true;
        try {
            for (int it = 0; it < round && more; ++it) {
                if (cap_nodes > 0 && S.popped >= cap_nodes) { more = false; break; }
                more = S.step(pq);
                if (replicated && pq.d.size() >= (size_t)(4 * world)) break;
            }
        } catch (const LpxException& e) {
            // a rank that left now would leave its peers waiting in the round's all-reduce: tell them, then stop together
            if (!sharded) throw;
            failed = true; fail_msg = e.what(); fail_code = e.code; more = false; replicated = false;
        }
        if (sharded && replicated && (!more || pq.d.empty())) {             // ended inside the replicated warm-up: one all-reduce says every rank did
            agree_h = fingerprint(false);
            replicated = false;
        }
        if (sharded && !replicated) {
            double vals[5] = {S.has_best ? S.best : -INFINITY, (more && !pq.d.empty()) ? 1.0 : 0.0, failed ? 1.0 : 0.0, -INFINITY, -INFINITY};
            const bool first = !agree_sent;
            if (first && !failed) { vals[3] = agree_h; vals[4] = -agree_h; }
            agree_sent = true;
            const double mine = vals[0];
            opt.allreduce_max(vals, 5);                                     // X1: incumbent + termination (+ "someone failed", + the fingerprint once)
            if (vals[0] > mine) { S.best = vals[0]; S.has_best = true; std::fill(S.bestX.begin(), S.bestX.end(), -1); }
            if (vals[2] > 0.0) { if (!failed) { failed = true; fail_code = LPX_EDEVICE; fail_msg = "sharded search: a peer rank failed"; } break; }
            if (first && vals[3] != -vals[4]) {
                failed = true; fail_code = LPX_EDEVICE;
                fail_msg = "sharded search: the replicated warm-up ended differently on different ranks (the bounds must be bit-identical across ranks)";
                break;
            }
            if (vals[1] == 0.0) break;
        } else if (!more || pq.d.empty()) {
            if (cap_nodes > 0 && S.popped >= cap_nodes) break;
            if (pq.d.empty()) break;
        }
    }
    if (failed) { try { S.finish_jobs(); } catch (const LpxException&) {} throw LpxException(fail_code ? fail_code : LPX_EDEVICE, fail_msg); }
    if (sharded) {
        double own = (S.has_best && !S.bestX.empty() && S.bestX[0] >= 0) ? -(double)rank : -INFINITY;
        opt.allreduce_max(&own, 1);
        std::vector<double> xs(n, -INFINITY);
        if (own == -(double)rank && S.has_best && S.bestX[0] >= 0) for (int i = 0; i < n; ++i) xs[i] = S.bestX[i];
        if (own != -INFINITY) { opt.allreduce_max(xs.data(), n); for (int i = 0; i < n; ++i) S.bestX[i] = (int32_t)xs[i]; }
    }

    S.finish_jobs();                                                        // a batch evaluated ahead may still be on the device
    if (const char* dbg = std::getenv("LPX_KNAP_DEBUG")) if (dbg[0] == '1')
        std::fprintf(stderr, "[lpx knap] sync launches %lld, async %lld, jobs %lld, begin %.1f ms, wait %.1f ms, device calls %.1f ms\n",
                     (long long)S.n_sync, (long long)S.n_async, (long long)S.jobs_run, S.begin_ms, S.wait_ms, S.dev_ms);

    // final report, :368-405 ("Report = finalReport, Summary = \"\"")
    std::string fr = "Final Report:\nBranch & Bound Knapsack Finished.\n\n";
    if (!S.has_best) fr += "Status: INFEASIBLE\n";
    else {
        fr += "Status: BEST CANDIDATE FOUND\n";
        for (int i = 0; i < n && n <= 4096; ++i) fr += "  x" + std::to_string(i + 1) + " = " + std::to_string(S.bestX[i]) + "\n";
        fr += "  z* = " + FormatNumber(S.best) + "\n";
    }
    fr += "\nSummary:\n";
    if (!S.has_best) fr += "No feasible candidate found.\n";
    else {
        fr += "Best Candidate = " + FormatNumber(S.best) + "\n";
        if (n <= 4096) { fr += "Best x* = ["; for (int i = 0; i < n; ++i) { if (i) fr += ", "; fr += std::to_string(S.bestX[i]); } fr += "]\n"; }
    }
    if (updatePivot) updatePivot(fr, nullptr);
    SimplexResult res;
    res.Report = fr; res.Summary = "";
    res.Status = S.has_best ? LPX_OPTIMAL : LPX_INFEASIBLE;
    // The reference returns text only; engine extras:
    res.OptimalValue = S.has_best ? S.best : -INFINITY;
    res.Solution.assign(S.bestX.begin(), S.bestX.end());
    res.HasSolution = false;
    // whole-job totals when summed over ranks: the replicated warm-up is counted by rank 0 only
    if (world > 1 && replicated && rank != 0) { S.redundant_popped = S.popped; S.redundant_relax = S.relaxations; }
    res.Nodes = S.popped - S.redundant_popped; res.LpSolves = S.relaxations - S.redundant_relax;
    res.NodeZ = {(double)(S.relaxations - S.redundant_relax), (double)(S.popped - S.redundant_popped), (double)S.expanded, (double)S.max_heap};
    res.Stats.launches = S.launches;
    res.Stats.loop_ms = S.dev_ms;                                           // time inside the device calls; the rest of the wall time is the host replay
    return res;
}

}}  // namespace lpx::host
