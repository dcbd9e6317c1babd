// host/bnb.cpp -- BranchAndBound mirror (Models/Branch&Bound.cs:20-304).
//
// Every node is an LP rebuilt from the root model plus its branching rows and re-solved from the
// slack basis on the GPU, exactly as the reference re-solves through LPSolver (:148).  Two modes:
//   faithful (bnb_mode 0)  DualSimplex keeps defects D1/D2: every `>=` child comes back without
//                          Solution/Tableau/Basis and is dropped as "Invalid" (:157-161).
//   repaired (bnb_mode 1)  DualSimplex runs with LPX_DUAL_REPAIRED; INFEASIBLE relaxations are pruned.
// Two searches:
//   bnb_search 0  the reference's recursive DFS, ceil child first (:256-257) -- node order, incumbent
//                 and node log are those of the reference.
//   bnb_search 1  level-synchronous frontier for the sharded node queue: the first levels are
//                 expanded redundantly on every rank until the frontier holds >= world*concurrent
//                 nodes, then frontier[i] belongs to rank i % world and stays local with its
//                 subtree; per level each rank solves its node LPs `concurrent_nodes` at a time
//                 (lpx_multi_run) and ONE all-reduce(max) over {incumbent z, have_work} is exchanged.
#include "model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <atomic>
#include <thread>

namespace lpx { namespace host {

namespace {

constexpr double EPS = 1e-6;      // :24
constexpr int MaxDepth = 200;     // :25

enum Outcome { O_ERROR = 0, O_INVALID = 1, O_INFEASIBLE_X = 2, O_PRUNED = 3, O_INCUMBENT = 4, O_NO_FRAC = 5,
               O_BRANCHED = 6, O_DEPTH = 7, O_LP_INFEASIBLE = 8 };

struct Cut { int var; Rel rel; double bound; };

// LPX_BNB_TIMING=1: print where the host spends its time (diagnostic)
struct PhaseTimer {
    double build = 0, run = 0, collect = 0, decide = 0, root = 0, total = 0, readback = 0, parking = 0; bool on = false;
    double drained = 0, drained_at = 0, drained_max = 0; int drains = 0;   // rolling batches: time with no window enqueued (the device idles once it has drained)
    PhaseTimer() { const char* e = std::getenv("LPX_BNB_TIMING"); on = e && e[0] == '1'; }
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    ~PhaseTimer() { if (on) std::fprintf(stderr, "[lpx bnb] root %.1f ms, build %.1f ms, run %.1f ms, collect %.1f ms (read-back %.1f, parking %.1f), decide %.1f ms, whole solves %.1f ms; no window in flight %d times, %.1f ms (longest %.1f)\n", root, build, run, collect, readback, parking, decide, total, drains, drained, drained_max); }
};
static PhaseTimer g_pt;

std::string last_error() { char b[1024]; lpx_last_error(b, sizeof(b)); return b; }

// pool of device tableaux keyed by shape: nodes of one depth share a shape.  Creating a handle costs ~4 ms (device and
// pinned allocations, a stream) and destroying it ~3 ms, so handles outlive the solve that made them: a finished pool
// parks them in a process-wide cache (bounded) and the next solve of the same shape class takes them from there.
struct HandleCache {
    std::map<std::pair<int, int>, std::vector<lpx_tableau*>> free_;
    size_t count = 0;
    static constexpr size_t kMax = 512;
    lpx_tableau* take(int R, int C) {
        auto it = free_.find({R, C});
        if (it == free_.end() || it->second.empty()) return nullptr;
        lpx_tableau* t = it->second.back(); it->second.pop_back(); --count;
        return t;
    }
    void park(lpx_tableau* t, int R, int C) {
        if (count >= kMax) { lpx_tableau_destroy(t); return; }
        free_[{R, C}].push_back(t); ++count;
    }
};
static HandleCache g_handle_cache;     // never destroyed: the HIP runtime may be gone by the time statics are torn down

struct HandlePool {
    std::map<std::pair<int, int>, std::vector<lpx_tableau*>> free_;
    std::vector<lpx_tableau*> all_;
    lpx_tableau* get(int R, int C) {
        auto& v = free_[{R, C}];
        if (!v.empty()) { lpx_tableau* t = v.back(); v.pop_back(); return t; }
        lpx_tableau* t = g_handle_cache.take(R, C);
        if (!t) {
            int rc = lpx_tableau_create(R, C, &t);
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
        }
        all_.push_back(t);
        cap_[t] = {R, C};
        return t;
    }
    std::map<lpx_tableau*, std::pair<int, int>> cap_;
    void put(lpx_tableau* t) { free_[cap_[t]].push_back(t); }
    ~HandlePool() { for (lpx_tableau* t : all_) g_handle_cache.park(t, cap_[t].first, cap_[t].second); }
};

struct NodeLP {
    // result of one relaxation
    bool error = false, has_solution = false;
    int status = LPX_OPTIMAL;
    std::vector<double> x; double z = 0.0;
    int64_t pivots = 0;
    // prepared tableau (host path: test seam) or branching-row descriptors (device assembly)
    std::vector<double> T; int R = 0, C = 0; std::vector<int32_t> basis; bool dual = false;
    bool on_device = false; std::vector<int32_t> cvar; std::vector<double> ccoef, czero, crhs;
    lpx_tableau* h = nullptr;
    // warm start: child of a parent whose final tableau is parked in a store slot
    lpx_store* pstore = nullptr; int pslot = -1; int prow = -1; bool warm = false; bool keep = false;   // keep: park the result
    lpx_store* kstore = nullptr; int kslot = -1; std::vector<int32_t> basis_out;
    double wbound = 0.0; bool wis_ge = false; int wslot = 0; int depth = 0;
    int feas = -1;          // IsFeasible(x) of :175-179 when it has been evaluated ahead of decide() (precompute_feas): 0 / 1
    const std::vector<Cut>* cuts_for_feas = nullptr;   // the node's branching rows, when solve_group may evaluate `feas` itself
};

// index of the root constraints for the per-node feasibility test (IsFeasibleIndexed below)
struct FeasIndex {
    bool built = false; int n = 0;
    std::vector<int> dense_rows; std::vector<double> ATd;          // [n][dense_rows.size()]
    struct Sparse { int row; std::vector<std::pair<int, double>> nz; };
    std::vector<Sparse> sparse;
    void build(const LPProblem& root)
    {
        n = root.NumVars();
        const int m = (int)root.Constraints.size();
        for (int r = 0; r < m; ++r) {
            const Constraint& c = root.Constraints[(size_t)r];
            int nnz = 0; for (int i = 0; i < n; ++i) if (c.A[(size_t)i] != 0.0) ++nnz;
            if (nnz * 8 > n) dense_rows.push_back(r);
            else { Sparse s; s.row = r; for (int i = 0; i < n; ++i) if (c.A[(size_t)i] != 0.0) s.nz.emplace_back(i, c.A[(size_t)i]); sparse.push_back(std::move(s)); }
        }
        const size_t nd = dense_rows.size();
        ATd.assign((size_t)n * nd, 0.0);
        for (size_t d = 0; d < nd; ++d) { const Constraint& c = root.Constraints[(size_t)dense_rows[d]]; for (int i = 0; i < n; ++i) ATd[(size_t)i * nd + d] = c.A[(size_t)i]; }
        built = true;
    }
};

struct Ctx {
    FeasIndex feas;
    const LPProblem* root; EngineOptions opt; UpdatePivot cb;
    double best = -INFINITY; bool has_best = false; std::vector<double> best_x;
    SimplexResult* out; HandlePool pool; bool stop = false;
    bool count_work = true;      // false while this rank only mirrors the replicated warm-up of rank 0
    // Node budget of the sharded searches.  It is evaluated IDENTICALLY on every rank while the warm-up is replicated
    // (budget_used counts every rank's copy of the same nodes), so all ranks leave that phase together; at the hand-out
    // what is left of the GLOBAL budget (opt.max_nodes) is split evenly and each rank then counts its own nodes.
    int64_t budget_used = 0, budget_cap = 0;
    bool over_budget() const { return budget_cap > 0 && budget_used >= budget_cap; }
    // DFS-order key of the incumbent's node: one bit per branching level, 0 = the ceil child the reference visits first
    // (Models/Branch&Bound.cs:256), 1 = the floor child.  Lexicographically smaller == earlier in the reference's DFS.
    std::vector<uint8_t> best_key; bool tie_by_key = false;
    // statistics of the sharded search (SimplexResult::Aux)
    int64_t levels = 0, allreduces = 0, rebalances = 0, moved = 0;
    std::vector<lpx_tableau*> released;
    std::map<std::pair<int, int>, lpx_store*> stores;
    lpx_store* store_for(lpx_tableau* h) {
        const std::pair<int, int> cap = pool.cap_[h];
        auto it = stores.find(cap);
        if (it != stores.end()) return it->second;
        lpx_store* s = nullptr;
        int rc = lpx_store_create(cap.first, cap.second, &s);
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        stores[cap] = s;
        return s;
    }
    // resident root templates: the prepared root tableau as the primal / dual path builds it
    lpx_tableau* root_tpl[2] = {nullptr, nullptr}; int tplR[2] = {0, 0}, tplC[2] = {0, 0}; bool tpl_bad[2] = {false, false};
    ~Ctx() { for (int w = 0; w < 2; ++w) release_exact_handle(root_tpl[w], tplR[w], tplC[w]); for (auto& kv : stores) lpx_store_destroy(kv.second); }
    void log(const std::string& s) { if (cb) cb(s + "\n", nullptr); }
};

LPProblem make_node(const LPProblem& root, const std::vector<Cut>& cuts)
{
    LPProblem p = root.Clone();                                   // :233, :242
    for (const Cut& c : cuts) {
        Constraint k; k.A.assign(root.NumVars(), 0.0); k.A[c.var] = 1.0;   // UnitVector, :298-303
        k.Relation = c.rel; k.B = c.bound;
        p.Constraints.push_back(k);
    }
    return p;
}

bool has_ge_or_eq(const LPProblem& p)                             // ChooseAlgorithm, :262-266
{
    for (const Constraint& c : p.Constraints) if (c.Relation == Rel::GE || c.Relation == Rel::EQ) return true;
    return false;
}

bool IsIntegral(const std::vector<double>& x)                     // :268-274
{
    for (double v : x) if (std::fabs(v - std::nearbyint(v)) > EPS) return false;
    return true;
}

// IsFeasible, :276-294, on the node = root constraints followed by its branching rows (same order as
// the cloned problem of the reference, without materialising the clone)
bool IsFeasible(const std::vector<double>& x, const LPProblem& root, const std::vector<Cut>& cuts)
{
    for (const Constraint& c : root.Constraints) {
        double sum = 0;
        for (size_t i = 0; i < x.size(); ++i) sum += c.A[i] * x[i];
        if (c.Relation == Rel::LE && sum > c.B + EPS) return false;
        if (c.Relation == Rel::GE && sum < c.B - EPS) return false;
        if (c.Relation == Rel::EQ && std::fabs(sum - c.B) > EPS) return false;
    }
    for (const Cut& k : cuts) {
        double sum = 0;                                     // UnitVector row: 0*x_i terms add exact zeros
        for (size_t i = 0; i < x.size(); ++i) sum += ((int)i == k.var ? 1.0 : 0.0) * x[i];
        if (k.rel == Rel::LE && sum > k.bound + EPS) return false;
        if (k.rel == Rel::GE && sum < k.bound - EPS) return false;
        if (k.rel == Rel::EQ && std::fabs(sum - k.bound) > EPS) return false;
    }
    for (double v : x) if (v < -EPS) return false;
    return true;
}

// The same test with the work laid out for the host's caches: identical sums, hence identical verdicts.
//  * a product with x_i == 0 is +-0 and leaves a running sum unchanged (the sums start at +0 and IEEE addition of a
//    zero never changes a value; the sign of a zero sum plays no part in the comparisons), so zero entries are skipped;
//  * rows with many coefficients are kept transposed ([variable][dense row]): the loop over the rows is contiguous and
//    every row still accumulates its terms in ascending variable order, one rounded multiply and one rounded add each
//    (the host code is built with -ffp-contract=off);
//  * rows with a few coefficients (the x_j <= 1 rows of a binary program) keep a list of their non-zeros;
//  * a branching row is a unit vector: its sum is x_k itself.
// Non-finite values fall back to the plain loop (0 * inf is not a zero).
bool row_ok(const Constraint& c, double sum)
{
    if (c.Relation == Rel::LE && sum > c.B + EPS) return false;
    if (c.Relation == Rel::GE && sum < c.B - EPS) return false;
    if (c.Relation == Rel::EQ && std::fabs(sum - c.B) > EPS) return false;
    return true;
}

bool IsFeasibleIndexed(const std::vector<double>& x, const LPProblem& root, const std::vector<Cut>& cuts, FeasIndex& ix)
{
    if (!ix.built) ix.build(root);
    for (double v : x) if (!std::isfinite(v)) return IsFeasible(x, root, cuts);
    for (const Constraint& c : root.Constraints) if ((int)c.A.size() < ix.n) return IsFeasible(x, root, cuts);
    const size_t nd = ix.dense_rows.size();
    std::vector<double> sums(nd, 0.0);
    for (int i = 0; i < ix.n && i < (int)x.size(); ++i) {
        const double xi = x[(size_t)i];
        if (xi == 0.0) continue;
        const double* col = ix.ATd.data() + (size_t)i * nd;
        for (size_t d = 0; d < nd; ++d) sums[d] += col[d] * xi;
    }
    for (size_t d = 0; d < nd; ++d) if (!row_ok(root.Constraints[(size_t)ix.dense_rows[d]], sums[d])) return false;
    for (const FeasIndex::Sparse& s : ix.sparse) {
        double sum = 0;
        for (const auto& e : s.nz) if ((size_t)e.first < x.size()) sum += e.second * x[(size_t)e.first];
        if (!row_ok(root.Constraints[(size_t)s.row], sum)) return false;
    }
    for (const Cut& k : cuts) {
        const double sum = 0.0 + x[(size_t)k.var];
        if (k.rel == Rel::LE && sum > k.bound + EPS) return false;
        if (k.rel == Rel::GE && sum < k.bound - EPS) return false;
        if (k.rel == Rel::EQ && std::fabs(sum - k.bound) > EPS) return false;
    }
    for (double v : x) if (v < -EPS) return false;
    return true;
}

// Prepares the relaxation exactly as PrimalSimplex.Solve / DualSimplex.Solve would (exceptions of
// the preparation become `error`, as BranchAndBound catches them, :150-154).
void prepare(const Ctx& c, const LPProblem& p, NodeLP& lp)
{
    std::vector<std::string> names;
    try {
        lp.dual = has_ge_or_eq(p);
        if (!lp.dual) {
            LPProblem model = p.Clone();
            if (model.ObjectiveSense == Sense::Min) for (double& v : model.C) v = -v;
            for (const Constraint& k : model.Constraints)
                if (k.B < -1e-9) { lp.error = true; return; }      // "negative RHS" exception, PrimalSimplex.cs:73-76
            BuildTableauPrimal(ExpandEqualitiesToInequalities(model), lp.T, lp.R, lp.C, lp.basis, names);
        } else {
            const bool repaired = c.opt.bnb_mode == 1;
            BuildTableauPrimal(PrepareForTableauDual(p, repaired), lp.T, lp.R, lp.C, lp.basis, names);
        }
    } catch (const LpxException&) { lp.error = true; }
}

// Device assembly: the root part of every node tableau is identical, so it is prepared and uploaded
// once per path (primal: ExpandEqualities + BuildTableau; dual: PrepareForTableau + BuildTableau) and a
// node only ships its branching rows.  Returns false when the root part itself throws in the reference
// (then every node of that path throws the same way).
bool ensure_template(Ctx& c, bool dual)
{
    const int w = dual ? 1 : 0;
    if (c.root_tpl[w] || c.tpl_bad[w]) return !c.tpl_bad[w];
    std::vector<double> T; int R, C; std::vector<int32_t> basis; std::vector<std::string> names;
    try {
        if (!dual) {
            LPProblem model = c.root->Clone();
            if (model.ObjectiveSense == Sense::Min) for (double& v : model.C) v = -v;
            for (const Constraint& k : model.Constraints) if (k.B < -1e-9) { c.tpl_bad[w] = true; return false; }
            BuildTableauPrimal(ExpandEqualitiesToInequalities(model), T, R, C, basis, names);
        } else {
            BuildTableauPrimal(PrepareForTableauDual(*c.root, c.opt.bnb_mode == 1), T, R, C, basis, names);
        }
    } catch (const LpxException&) { c.tpl_bad[w] = true; return false; }
    c.root_tpl[w] = acquire_exact_handle(R, C);
    c.tplR[w] = R; c.tplC[w] = C;
    int rc = lpx_tableau_upload(c.root_tpl[w], T.data(), basis.data());
    if (rc) throw LpxException(rc, "liblpx: " + last_error());
    c.tplR[w] = R; c.tplC[w] = C;
    return true;
}

// branching rows exactly as PrimalSimplex.Solve / PrepareForTableau would leave them
void prepare_device(Ctx& c, const std::vector<Cut>& cuts, NodeLP& lp)
{
    lp.dual = has_ge_or_eq(*c.root);
    for (const Cut& k : cuts) if (k.rel != Rel::LE) lp.dual = true;
    if (!ensure_template(c, lp.dual)) { lp.error = true; return; }
    const int w = lp.dual ? 1 : 0;
    const bool fix_d1 = c.opt.bnb_mode == 1;
    for (const Cut& k : cuts) {
        double a = 1.0, z = 0.0, B = k.bound;
        if (!lp.dual) {
            if (B < -1e-9) { lp.error = true; return; }              // "negative RHS" exception, PrimalSimplex.cs:73-76
        } else {
            if (k.rel == Rel::GE) { a *= -1; z *= -1; B *= -1; }      // DualSimplex.cs:141-147
            if (!fix_d1 && B < -1e-9) { a *= -1; z *= -1; B *= -1; }  // :148-153 (defect D1)
        }
        lp.cvar.push_back(k.var); lp.ccoef.push_back(a); lp.czero.push_back(z); lp.crhs.push_back(B);
    }
    lp.R = c.tplR[w] + (int)cuts.size(); lp.C = c.tplC[w] + (int)cuts.size();
    lp.on_device = true;
}

int cap_slack(int d) { const int s = (d + 31) / 32 * 32; return s > 0 ? s : 32; }

void upload(Ctx& c, NodeLP& lp)
{
    if (lp.warm) {
        const int d = lp.depth, w = lp.wslot;
        lp.h = c.pool.get(c.tplR[w] + cap_slack(d), c.tplC[w] + cap_slack(d));
        int rc = lpx_tableau_build_child_from_store(lp.h, lp.pstore, lp.pslot, lp.cvar.back(), lp.prow, lp.wis_ge ? 1 : 0, lp.wbound);
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        return;
    }
    if (lp.on_device) {
        // capacity classes of 32 levels: one set of handles (and captured graphs) serves 32 depths
        const int d = (int)lp.cvar.size(), w = lp.dual ? 1 : 0;
        const int slack = cap_slack(d);
        lp.h = c.pool.get(c.tplR[w] + slack, c.tplC[w] + slack);
        int rc = lpx_tableau_build_node(lp.h, c.root_tpl[lp.dual ? 1 : 0], (int)lp.cvar.size(), lp.cvar.data(),
                                        lp.ccoef.data(), lp.czero.data(), lp.crhs.data());
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        return;
    }
    lp.h = c.pool.get(lp.R, lp.C);
    int rc = lpx_tableau_upload(lp.h, lp.T.data(), lp.basis.data());
    if (rc) throw LpxException(rc, "liblpx: " + last_error());
    std::vector<double>().swap(lp.T);
}

// Results of a whole batch: one launch + one wait reads every node's solution, the final tableaux that stay as
// parents are parked with all their copies in flight together (a per-node form waits twice per node).
void collect_group(Ctx& c, std::vector<NodeLP*>& live, const std::vector<int>& st, const std::vector<lpx_stats>& ss, int nvars)
{
    std::vector<NodeLP*> want; std::vector<lpx_tableau*> hs; int maxm = 1;
    for (size_t i = 0; i < live.size(); ++i) {
        NodeLP& lp = *live[i];
        lp.pivots = ss[i].pivots;
        if (c.count_work) { c.out->Stats.pivots += ss[i].pivots; c.out->Stats.launches += ss[i].launches; c.out->Stats.loop_ms += ss[i].loop_ms; }
        if (st[i] < 0) throw LpxException(st[i], "liblpx: " + last_error());
        lp.status = st[i];
        if (st[i] == LPX_ITER_LIMIT) lp.error = true;                    // exception in the reference
        else if (lp.dual && c.opt.bnb_mode == 0) lp.has_solution = false;   // defect D2
        else { want.push_back(&lp); hs.push_back(lp.h); maxm = std::max(maxm, lp.R - 1); }
    }
    if (!want.empty()) {
        std::vector<double> xs((size_t)nvars * want.size()), zs(want.size());
        std::vector<int32_t> bs((size_t)maxm * want.size());
        double t0 = PhaseTimer::now();
        int rc = lpx_multi_solution(hs.data(), (int)hs.size(), nvars, xs.data(), zs.data(), bs.data(), maxm);
        g_pt.readback += PhaseTimer::now() - t0;
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        std::vector<lpx_store*> stores; std::vector<lpx_tableau*> keep_h; std::vector<NodeLP*> keep_lp;
        for (size_t i = 0; i < want.size(); ++i) {
            NodeLP& lp = *want[i];
            lp.x.assign(xs.begin() + (size_t)i * nvars, xs.begin() + (size_t)(i + 1) * nvars);
            lp.z = zs[i];
            lp.basis_out.assign(std::max(lp.R - 1, 1), 0);
            if (lp.keep && lp.R > 1) std::copy(bs.begin() + (size_t)i * maxm, bs.begin() + (size_t)i * maxm + (lp.R - 1), lp.basis_out.begin());
            lp.has_solution = true;
            if (lp.keep && lp.status == LPX_OPTIMAL) { stores.push_back(c.store_for(lp.h)); keep_h.push_back(lp.h); keep_lp.push_back(&lp); }
        }
        if (!keep_h.empty()) {                                           // park the final tableaux for the children
            std::vector<int> slots(keep_h.size(), -1);
            t0 = PhaseTimer::now();
            rc = lpx_store_save_multi(stores.data(), keep_h.data(), (int)keep_h.size(), slots.data());
            g_pt.parking += PhaseTimer::now() - t0;
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
            for (size_t i = 0; i < keep_lp.size(); ++i) { keep_lp[i]->kstore = stores[i]; keep_lp[i]->kslot = slots[i]; }
        }
    }
    for (NodeLP* lp : live) { c.pool.put(lp->h); lp->h = nullptr; }
}

void solve_group(Ctx& c, std::vector<NodeLP*>& group, int nvars)
{
    lpx_run_opts po, dopt; lpx_default_opts(&po, 0); lpx_default_opts(&dopt, 1);
    po.max_iter = dopt.max_iter = c.opt.max_iter; po.batch = dopt.batch = c.opt.batch;
    if (c.opt.bnb_mode == 1) { dopt.fdf_guard = c.opt.max_iter; dopt.cleanup = 1; }
    if (c.opt.test_fail_after_nodes > 0 && c.budget_used >= c.opt.test_fail_after_nodes && !group.empty())
        throw LpxException(LPX_ENOMEM, "test seam: injected failure of a node group");       // include/lpx_test.h
    bool any_warm = false; for (NodeLP* lp : group) if (lp->warm) any_warm = true;
    bool rolling = false;
    if (any_warm) {                                               // dual feasible start: only the dual loop (and its clean-up) runs
        dopt.fdf_guard = 0; dopt.cleanup = 1;
        // A warm-started node needs a few dozen pivots.  Until r03 the big ones (config 4: 8 MB) streamed 64 at a time through the
        // one-launch-per-step kernel (15.8 MB of HBM traffic per node and pivot), because only four fitted the chip.  Twelve fit now
        // (rows in registers + LDS), a whole group goes out as one launch and the hardware starts the next node as one leaves: the
        // resident loop takes 0.45 ms per node whatever the HBM does -- 11.8 k against 8.2 k nodes/s on the same box, node logs
        // and node objective values bitwise the same.  Streaming rolling batches remain for nodes wider than the register
        // kernel's 1536 columns (and behind LPX_WARM_RESIDENT=0).
        size_t node_bytes = 0; int node_cols = 0;
        for (NodeLP* lp : group) { node_bytes = std::max(node_bytes, sizeof(double) * (size_t)lp->R * (size_t)lp->C); node_cols = std::max(node_cols, lp->C); }
        const bool warm_stream = [] { const char* e = std::getenv("LPX_WARM_RESIDENT"); return e && e[0] == '0'; }();   // read per solve (bench.py times both forms in one process)
        if ((warm_stream || node_cols > 1536) && node_bytes > ((size_t)1 << 20)) {
            dopt.resident = -1; po.resident = -1; rolling = group.size() > 1;
            static const int roll_batch = [] { const char* e = std::getenv("LPX_ROLL_BATCH"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 16; }();   // pivots between polls
            if (rolling && c.opt.batch <= 0) dopt.batch = po.batch = roll_batch;
        }
    }
    if (c.opt.test_node_lp) {          // test seam (include/lpx.h): the device loop is stood in for
        for (NodeLP* lp : group) {
            if (c.count_work) c.out->LpSolves++;
            if (lp->error || lp->R < 2) { lp->error = true; continue; }
            lp->x.assign(nvars, 0.0);
            int64_t piv = 0;
            int st = c.opt.test_node_lp(lp->T.data(), lp->R, lp->C, lp->basis.data(), lp->dual ? 1 : 0, c.opt.bnb_mode,
                                        c.opt.max_iter, nvars, lp->x.data(), &lp->z, &piv);
            c.out->Stats.pivots += piv; lp->pivots = piv; lp->status = st;
            if (st < 0) throw LpxException(st, "test seam: node LP failed");          // as collect_group does for lpx_multi_run
            if (st == LPX_ITER_LIMIT) lp->error = true;
            else lp->has_solution = !(lp->dual && c.opt.bnb_mode == 0);
        }
        return;
    }
    const size_t width = (size_t)std::max(1, c.opt.concurrent_nodes);
    // gives every node of `nodes` a handle and builds its tableau on the device (one launch per kind); returns the admitted ones
    auto admit = [&](const std::vector<NodeLP*>& nodes, std::vector<NodeLP*>& live) {
        const double t0 = PhaseTimer::now();
        std::vector<NodeLP*> assemble[2];                     // cold nodes built on the device, per root template: ONE launch each
        std::vector<NodeLP*> warm_kids;                       // warm-started children, built from their parked parents: ONE launch
        for (NodeLP* lp : nodes) {
            if (lp->error || lp->R < 2) { if (lp->R < 2) lp->error = true; continue; }
            if (lp->warm) {
                const int d = lp->depth, w = lp->wslot;
                lp->h = c.pool.get(c.tplR[w] + cap_slack(d), c.tplC[w] + cap_slack(d));
                warm_kids.push_back(lp);
            } else if (lp->on_device) {
                const int d = (int)lp->cvar.size(), w = lp->dual ? 1 : 0;
                lp->h = c.pool.get(c.tplR[w] + cap_slack(d), c.tplC[w] + cap_slack(d));   // capacity classes of 32 levels (see upload)
                assemble[w].push_back(lp);
            } else upload(c, *lp);
            live.push_back(lp);
        }
        for (int w = 0; w < 2; ++w) {
            if (assemble[w].empty()) continue;
            std::vector<lpx_tableau*> nh; std::vector<int32_t> off{0}, var; std::vector<double> coef, zero, rhs;
            for (NodeLP* lp : assemble[w]) {
                nh.push_back(lp->h);
                var.insert(var.end(), lp->cvar.begin(), lp->cvar.end()); coef.insert(coef.end(), lp->ccoef.begin(), lp->ccoef.end());
                zero.insert(zero.end(), lp->czero.begin(), lp->czero.end()); rhs.insert(rhs.end(), lp->crhs.begin(), lp->crhs.end());
                off.push_back((int32_t)var.size());
            }
            int rc = lpx_tableau_build_nodes(nh.data(), c.root_tpl[w], (int)nh.size(), off.data(), var.data(), coef.data(), zero.data(), rhs.data());
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
        }
        if (!warm_kids.empty()) {
            std::vector<lpx_tableau*> ch; std::vector<lpx_store*> st; std::vector<int> slots; std::vector<int32_t> var, row, ge; std::vector<double> bd;
            for (NodeLP* lp : warm_kids) {
                ch.push_back(lp->h); st.push_back(lp->pstore); slots.push_back(lp->pslot);
                var.push_back(lp->cvar.back()); row.push_back(lp->prow); ge.push_back(lp->wis_ge ? 1 : 0); bd.push_back(lp->wbound);
            }
            int rc = lpx_tableau_build_children_from_store(ch.data(), st.data(), slots.data(), (int)ch.size(), var.data(), row.data(), ge.data(), bd.data());
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
        }
        g_pt.build += PhaseTimer::now() - t0;
    };
    if (rolling) {
        // ROLLING batches (warm-started children on the streaming kernels): most of them need a few dozen pivots, a few need hundreds.
        // TWO batches of `width` nodes alternate (lpx_multi_run_begin / _end): while one pivots through a window of `roll_batch`
        // steps, the host reads the other one's finished nodes back, parks their tableaux, tests their solutions and assembles fresh
        // children in their places -- so the device never waits for the host and finished nodes leave the grid after every window.
        static const bool pipelined = [] { const char* e = std::getenv("LPX_ROLL_PIPELINE"); return !(e && e[0] == '0'); }();   // diagnostic: 0 = one batch, synchronous
        size_t next = 0;
        const int steps = dopt.batch > 0 ? dopt.batch : 16;
        struct Set { std::vector<NodeLP*> inflight; bool running = false; };
        static const int NS = [] { const char* e = std::getenv("LPX_ROLL_SETS"); const int v = e ? std::atoi(e) : 0; return v >= 2 && v <= LPX_ASYNC_SLOTS ? v : 2; }();
        Set sets[LPX_ASYNC_SLOTS];
        auto others_running = [&](int s) { for (int k = 0; k < NS; ++k) if (k != s && sets[k].running) return true; return false; };
        bool async_ok = pipelined;
        auto launch = [&](int s) {
            Set& S = sets[s];
            std::vector<NodeLP*> fresh;
            while (next < group.size() && S.inflight.size() + fresh.size() < width) fresh.push_back(group[next++]);
            if (c.count_work) c.out->LpSolves += (int64_t)fresh.size();
            admit(fresh, S.inflight);
            if (S.inflight.empty()) return;
            std::vector<lpx_tableau*> hs; std::vector<int> dual;
            for (NodeLP* lp : S.inflight) { hs.push_back(lp->h); dual.push_back(lp->dual ? 1 : 0); }
            const int rc = lpx_multi_run_begin(s, hs.data(), dual.data(), (int)hs.size(), &po, &dopt, steps);
            if (rc < 0) throw LpxException(rc, "liblpx: " + last_error());
            if (rc == 1) { async_ok = false; return; }               // not available: the synchronous loop below takes what is in the sets
            if (g_pt.on && g_pt.drained_at > 0 && !others_running(s)) {
                const double d = PhaseTimer::now() - g_pt.drained_at;
                g_pt.drained += d; g_pt.drains++; g_pt.drained_max = std::max(g_pt.drained_max, d);
            }
            S.running = true;
        };
        auto land = [&](int s) {
            Set& S = sets[s];
            std::vector<int> st(S.inflight.size()); std::vector<lpx_stats> ss(S.inflight.size());
            double t0 = PhaseTimer::now();
            const int rc = lpx_multi_run_end(s, st.data(), ss.data());
            g_pt.run += PhaseTimer::now() - t0;
            S.running = false;
            if (g_pt.on && !others_running(s)) g_pt.drained_at = PhaseTimer::now();
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
            std::vector<NodeLP*> fin, keep; std::vector<int> fst; std::vector<lpx_stats> fss;
            for (size_t i = 0; i < S.inflight.size(); ++i) {
                if (st[i] == LPX_RUNNING) keep.push_back(S.inflight[i]);
                else { fin.push_back(S.inflight[i]); fst.push_back(st[i]); fss.push_back(ss[i]); }
            }
            t0 = PhaseTimer::now();
            collect_group(c, fin, fst, fss, nvars);                  // pivots are cumulative per node: counted once, when it finishes
            // the feasibility test of :175-179 for the finished nodes, here where the other batch's window hides it
            if (c.feas.built || !fin.empty()) {
                if (!c.feas.built) c.feas.build(*c.root);
                for (NodeLP* lp : fin)
                    if (lp->cuts_for_feas && !lp->error && lp->has_solution && !(c.opt.bnb_mode == 1 && lp->status == LPX_INFEASIBLE))
                        lp->feas = IsFeasibleIndexed(lp->x, *c.root, *lp->cuts_for_feas, c.feas) ? 1 : 0;
            }
            g_pt.collect += PhaseTimer::now() - t0;
            S.inflight.swap(keep);
        };
        if (async_ok) {
            try {
                for (int s = 0; s < NS && async_ok; ++s) launch(s);
                while (async_ok && others_running(-1))
                    for (int s = 0; s < NS && async_ok; ++s) {
                        if (!sets[s].running) continue;
                        land(s);
                        launch(s);
                    }
            } catch (...) {
                // a window may still be in flight in the other slot: wait for it before the handles go back to the pool
                for (int s = 0; s < NS; ++s) if (sets[s].running) { std::vector<int> st(sets[s].inflight.size()); lpx_multi_run_end(s, st.data(), nullptr); sets[s].running = false; }
                throw;
            }
            if (async_ok) return;
            for (int s = 0; s < NS; ++s) if (sets[s].running) land(s);
        }
        // synchronous form (one batch; also what is left when the two-slot path was not available)
        std::vector<NodeLP*> inflight;
        for (int s = 0; s < 2; ++s) { inflight.insert(inflight.end(), sets[s].inflight.begin(), sets[s].inflight.end()); sets[s].inflight.clear(); }
        while (next < group.size() || !inflight.empty()) {
            std::vector<NodeLP*> fresh;
            while (next < group.size() && inflight.size() + fresh.size() < width) fresh.push_back(group[next++]);
            if (c.count_work) c.out->LpSolves += (int64_t)fresh.size();
            admit(fresh, inflight);
            if (inflight.empty()) continue;
            std::vector<lpx_tableau*> hs; std::vector<int> dual;
            for (NodeLP* lp : inflight) { hs.push_back(lp->h); dual.push_back(lp->dual ? 1 : 0); }
            std::vector<int> st(hs.size()); std::vector<lpx_stats> ss(hs.size());
            static const int roll_div = [] { const char* e = std::getenv("LPX_ROLL_DIV"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 2; }();   // diagnostic
            const int min_active = next < group.size() ? (int)std::max<size_t>(1, width / roll_div) : 0;
            double t0 = PhaseTimer::now();
            int rc = lpx_multi_run_some(hs.data(), dual.data(), (int)hs.size(), &po, &dopt, st.data(), ss.data(), min_active);
            g_pt.run += PhaseTimer::now() - t0;
            if (rc) throw LpxException(rc, "liblpx: " + last_error());
            std::vector<NodeLP*> fin, keep; std::vector<int> fst; std::vector<lpx_stats> fss;
            for (size_t i = 0; i < inflight.size(); ++i) {
                if (st[i] == LPX_RUNNING) keep.push_back(inflight[i]);
                else { fin.push_back(inflight[i]); fst.push_back(st[i]); fss.push_back(ss[i]); }
            }
            // pivots are cumulative per node: count a node's once, when it finishes
            t0 = PhaseTimer::now();
            collect_group(c, fin, fst, fss, nvars);
            g_pt.collect += PhaseTimer::now() - t0;
            inflight.swap(keep);
        }
        return;
    }
    for (size_t a = 0; a < group.size(); a += width) {
        const size_t b = std::min(group.size(), a + width);
        std::vector<NodeLP*> live;
        admit(std::vector<NodeLP*>(group.begin() + a, group.begin() + b), live);
        if (c.count_work) c.out->LpSolves += (int64_t)(b - a);        // every node reaches _solver.Solve (:148), even if it throws
        if (live.empty()) continue;
        std::vector<lpx_tableau*> hs; std::vector<int> dual;
        for (NodeLP* lp : live) { hs.push_back(lp->h); dual.push_back(lp->dual ? 1 : 0); }
        std::vector<int> st(hs.size()); std::vector<lpx_stats> ss(hs.size());
        double t0 = PhaseTimer::now();
        int rc = lpx_multi_run(hs.data(), dual.data(), (int)hs.size(), &po, &dopt, st.data(), ss.data());
        g_pt.run += PhaseTimer::now() - t0;
        if (rc) throw LpxException(rc, "liblpx: " + last_error());
        t0 = PhaseTimer::now();
        collect_group(c, live, st, ss, nvars);
        g_pt.collect += PhaseTimer::now() - t0;
    }
}

// The feasibility test of :175-179 for a whole level ahead of the decisions: it is a pure function of a node's x (768 rows x 512
// columns at config 4: ~15 us per node, a tenth of the warm-started leg when done one node at a time between the device
// batches), so the nodes of a level are tested on a few host threads; decide() then takes the verdict from NodeLP::feas.  Order
// of the decisions, logs and incumbent updates is untouched.  LPX_HOST_THREADS=n (default: up to 8; 1 = off).
template <class FN>
void precompute_feas(Ctx& c, std::vector<NodeLP>& lps, const std::vector<FN>& frontier, const std::vector<char>& skip)
{
    static const int nthreads = [] {
        const char* e = std::getenv("LPX_HOST_THREADS"); int v = e ? std::atoi(e) : 0;
        if (v <= 0) { v = (int)std::thread::hardware_concurrency(); if (v > 8) v = 8; }
        return v < 1 ? 1 : v; }();
    std::vector<size_t> todo;
    for (size_t i = 0; i < lps.size(); ++i) {
        const NodeLP& lp = lps[i];
        if (skip[i] || lp.error || !lp.has_solution || lp.feas >= 0 || (c.opt.bnb_mode == 1 && lp.status == LPX_INFEASIBLE)) continue;
        todo.push_back(i);
    }
    if (nthreads < 2 || todo.size() < 64) return;
    if (!c.feas.built) c.feas.build(*c.root);
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= todo.size()) break;
            const size_t i = todo[k];
            lps[i].feas = IsFeasibleIndexed(lps[i].x, *c.root, frontier[i].cuts, c.feas) ? 1 : 0;
        }
    };
    std::vector<std::thread> pool;
    const int n = std::min<int>(nthreads, (int)(todo.size() / 32) + 1);
    for (int t = 1; t < n; ++t) pool.emplace_back(work);
    work();
    for (std::thread& th : pool) th.join();
}

void node_log(Ctx& c, int depth, int outcome, int var, double z)
{
    c.out->NodeLog.push_back(depth); c.out->NodeLog.push_back(outcome); c.out->NodeLog.push_back(var);
    c.out->NodeZ.push_back(z);
}

std::vector<uint8_t> key_of(const std::vector<Cut>& cuts)
{
    std::vector<uint8_t> k(cuts.size());
    for (size_t i = 0; i < cuts.size(); ++i) k[i] = cuts[i].rel == Rel::GE ? 0 : 1;
    return k;
}

void set_incumbent(Ctx& c, const std::vector<double>& x, double z, const std::vector<Cut>& cuts = {})
{
    c.best = z; c.has_best = true;
    c.best_x.resize(x.size());
    for (size_t i = 0; i < x.size(); ++i) c.best_x[i] = std::nearbyint(x[i]);    // RoundInt, :296
    c.best_key = key_of(cuts);
}

// The decision part of SolveNode (:156-230) for a solved relaxation.  Returns the branching
// variable (>= 0) when the node branches, else -1.
int decide(Ctx& c, const std::vector<Cut>& cuts, const NodeLP& lp, int depth, const std::string& name, int& floorVal, int& ceilVal)
{
    if (lp.error) { c.log(name + ": LP relaxation infeasible or error"); node_log(c, depth, O_ERROR, -1, 0.0); return -1; }
    if (!lp.has_solution) {
        c.log(name + ": Invalid Simplex result (missing Solution, Tableau, Basis, or VarNames).");
        node_log(c, depth, O_INVALID, -1, 0.0); return -1;
    }
    if (c.opt.bnb_mode == 1 && lp.status == LPX_INFEASIBLE) { node_log(c, depth, O_LP_INFEASIBLE, -1, lp.z); return -1; }
    const std::vector<double>& x = lp.x; const double z = lp.z;
    if (c.cb) c.log(name + " LP solution: z* = " + FormatF(z, 3));
    if (!(lp.feas >= 0 ? lp.feas == 1 : IsFeasibleIndexed(x, *c.root, cuts, c.feas))) { c.log(name + ": Solution is infeasible for constraints."); node_log(c, depth, O_INFEASIBLE_X, -1, z); return -1; }   // :175-179
    const double bestObj = c.has_best ? c.best : -INFINITY;
    // Sharded searches visit nodes level by level, not in the reference's depth-first order.  Among integer nodes with
    // EXACTLY the incumbent's z the one the reference's DFS reaches first keeps the solution vector (first found wins
    // there, :182-195): smaller DFS-order key.  Counters and the node log are untouched (the node is pruned as always).
    if (c.tie_by_key && c.has_best && z == c.best && IsIntegral(x)) {
        const std::vector<uint8_t> k = key_of(cuts);
        if (c.best_x.empty() || k < c.best_key) {
            c.best_x.resize(x.size());
            for (size_t i = 0; i < x.size(); ++i) c.best_x[i] = std::nearbyint(x[i]);
            c.best_key = k;
        }
    }
    if (z <= bestObj + EPS) { c.log(name + ": Pruned by bound (z* <= current best " + FormatF(bestObj, 3) + ")."); node_log(c, depth, O_PRUNED, -1, z); return -1; }   // :182-186
    if (IsIntegral(x)) {                                                     // :189-195
        set_incumbent(c, x, z, cuts);
        c.log(name + " is integer feasible. Updated BestObjective = " + FormatF(z, 3));
        node_log(c, depth, O_INCUMBENT, -1, z); return -1;
    }
    int fracIndex = -1; double minDist = 1.7976931348623157e308;             // :198-213
    for (int i = 0; i < (int)x.size(); ++i) {
        double fracPart = x[i] - std::floor(x[i]);
        if (fracPart > EPS && (1 - fracPart) > EPS) {
            double dist = std::fabs(fracPart - 0.5);
            if (dist < minDist || (dist == minDist && i < fracIndex)) { minDist = dist; fracIndex = i; }
        }
    }
    if (fracIndex == -1) { node_log(c, depth, O_NO_FRAC, -1, z); return -1; }  // :215-219
    floorVal = (int)std::floor(x[fracIndex]);                                // :222-223
    ceilVal = (int)std::ceil(x[fracIndex]);
    c.log(name + ": Branching on x" + std::to_string(fracIndex + 1) + " = " + FormatF(x[fracIndex], 3) +
          " (floor=" + std::to_string(floorVal) + ", ceil=" + std::to_string(ceilVal) + ")");
    node_log(c, depth, O_BRANCHED, fracIndex, z);
    return fracIndex;
}

// ---- search 0: the reference's recursion (:128-258) ---------------------------------------------------
void SolveNode(Ctx& c, std::vector<Cut>& cuts, int depth, const std::string& name)
{
    if (c.stop) return;
    if (c.opt.max_nodes > 0 && c.out->Nodes >= c.opt.max_nodes) { c.stop = true; return; }
    c.out->Nodes++;
    if (depth > MaxDepth) { c.log(name + ": Maximum recursion depth reached -> prune."); node_log(c, depth, O_DEPTH, -1, 0.0); return; }
    NodeLP lp;
    if (c.opt.test_node_lp) prepare(c, make_node(*c.root, cuts), lp); else prepare_device(c, cuts, lp);
    std::vector<NodeLP*> g{&lp};
    solve_group(c, g, c.root->NumVars());
    int fl = 0, ce = 0;
    int k = decide(c, cuts, lp, depth, name, fl, ce);
    if (k < 0) return;
    cuts.push_back({k, Rel::GE, (double)ce});
    SolveNode(c, cuts, depth + 1, "Subproblem: x" + std::to_string(k + 1) + " >= " + std::to_string(ce));   // ceil first, :256
    cuts.back() = {k, Rel::LE, (double)fl};
    SolveNode(c, cuts, depth + 1, "Subproblem: x" + std::to_string(k + 1) + " <= " + std::to_string(fl));   // :257
    cuts.pop_back();
}

// ---- search 1: level-synchronous frontier -----------------------------------------------------------
struct FNode { std::vector<Cut> cuts; int depth; };

// Everything the ranks tell each other goes through the ONE collective the boundary knows, all-reduce(MAX) in place
// (RCCL in production): a value only rank r knows travels in slot r of a vector the other ranks fill with -inf.
struct Exchange {
    Ctx& c; int world, rank;
    void max(double* v, int n) { c.opt.allreduce_max(v, n); c.allreduces++; }
};

// final ownership of the solution vector: among the ranks holding an x for the global best z, the one whose node comes
// first in the reference's DFS order (smallest key) publishes it -- ONE all-reduce for the keys, one for x
void publish_best(Ctx& c, Exchange& ex, int nvars)
{
    constexpr int CH = 5, BITS = 48;                                   // 240 key bits >= MaxDepth + 1 levels
    const int world = ex.world, rank = ex.rank;
    std::vector<double> keys((size_t)world * (CH + 1), -INFINITY);
    const bool cand = c.has_best && !c.best_x.empty();
    if (cand) {
        keys[(size_t)rank * (CH + 1)] = 1.0;                           // "rank holds an x for the best z"
        for (int k = 0; k < CH; ++k) {
            double v = 0.0;
            for (int b = 0; b < BITS; ++b) { const size_t i = (size_t)k * BITS + b; v = v * 2.0 + (i < c.best_key.size() ? c.best_key[i] : 1); }
            keys[(size_t)rank * (CH + 1) + 1 + k] = v;
        }
    }
    ex.max(keys.data(), (int)keys.size());
    int owner = -1;
    for (int r = 0; r < world; ++r) {
        if (keys[(size_t)r * (CH + 1)] != 1.0) continue;
        if (owner < 0) { owner = r; continue; }
        for (int k = 1; k <= CH; ++k) {
            const double a = keys[(size_t)r * (CH + 1) + k], b = keys[(size_t)owner * (CH + 1) + k];
            if (a != b) { if (a < b) owner = r; break; }
        }
    }
    if (owner < 0) return;
    std::vector<double> xs(nvars, -INFINITY);
    if (owner == rank) xs = c.best_x;
    ex.max(xs.data(), nvars);
    c.best_x = xs;
}

// The replicated warm-up relies on every rank computing the SAME frontier (same LPs, same arithmetic, same decisions).  If that ever
// fails -- round 2's record scale7: a kernel whose result depended on workgroup scheduling under GPU sharing gave the ranks different
// root LPs; they left the replicated phase on different levels and then sat in collectives that no longer matched until the job was
// killed -- the ranks must find out instead of waiting for good: a fingerprint of the frontier as it stands when the replicated
// phase ends (hand-out, or the search running dry / out of budget before it) travels as {+h, -h} in the first all-reduce of every
// rank; max(+h) == -max(-h) iff all ranks agree.  Every rank of a sharded search issues that first all-reduce, also one whose search
// ended while replicated.  52 bits: exact in a double.
template <class FN> double frontier_fingerprint(const std::vector<FN>& frontier, bool handed_out)
{
    uint64_t h = 1469598103934665603ull ^ (handed_out ? 0x9e3779b97f4a7c15ull : 0ull);
    auto mix = [&](uint64_t v) { h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2); };
    mix((uint64_t)frontier.size());
    for (const FN& f : frontier) {
        mix((uint64_t)f.cuts.size());
        for (const Cut& k : f.cuts) { mix((uint64_t)(uint32_t)k.var); mix((uint64_t)(int)k.rel); int64_t b; double d = k.bound; std::memcpy(&b, &d, sizeof(b)); mix((uint64_t)b); }
    }
    return (double)(h & ((1ull << 52) - 1));
}

void LevelSearch(Ctx& c)
{
    const int world = std::max(1, c.opt.world), rank = c.opt.rank;
    const bool sharded = (world > 1 || c.opt.shard_one) && (bool)c.opt.allreduce_max;
    Exchange ex{c, world, rank};
    c.tie_by_key = true;
    // The replicated warm-up only has to seed every rank with a subtree: 2 nodes per rank.  Its solves are
    // redundant across ranks, so only rank 0 counts them (LpSolves / Nodes stay whole-job totals when summed).
    const size_t want = (size_t)world * 2;
    std::vector<FNode> frontier{FNode{{}, 0}};
    bool replicated = world > 1;
    c.count_work = !(replicated && rank != 0);
    c.budget_used = 0; c.budget_cap = c.opt.max_nodes;
    bool failed = false; std::string fail_msg; int fail_code = 0;
    double agree_h = -1.0; bool agree_sent = !sharded;      // fingerprint of the end of the replicated phase, sent once
    if (sharded && !replicated) agree_h = frontier_fingerprint(frontier, false);       // a world of one (LPX_COMM_SHARD_ONE): nothing was replicated
    for (;;) {
        if (replicated && frontier.size() >= want) {
            // hand the replicated frontier out: node i -> rank i % world, subtrees stay local from here on
            agree_h = frontier_fingerprint(frontier, true);
            std::vector<FNode> mine;
            for (size_t i = 0; i < frontier.size(); ++i) if ((int)(i % world) == rank) mine.push_back(std::move(frontier[i]));
            frontier.swap(mine);
            replicated = false;
            c.count_work = true;
            if (c.opt.max_nodes > 0) {                                 // the rest of the global budget, split evenly
                const int64_t left = std::max<int64_t>(0, c.opt.max_nodes - c.budget_used);
                c.budget_cap = c.budget_used + (left + world - 1) / world;
            }
        }
        // this round's nodes: the whole frontier (breadth first) or, with bnb_dive, its deepest K nodes
        std::vector<FNode> parked;
        if (c.opt.bnb_dive && !replicated && frontier.size() > (size_t)std::max(1, c.opt.concurrent_nodes)) {
            std::stable_sort(frontier.begin(), frontier.end(), [](const FNode& a, const FNode& b) { return a.depth > b.depth; });
            const size_t K = (size_t)std::max(1, c.opt.concurrent_nodes);
            parked.assign(std::make_move_iterator(frontier.begin() + K), std::make_move_iterator(frontier.end()));
            frontier.resize(K);
        }
        // solve this level
        std::vector<NodeLP> lps(frontier.size());
        std::vector<NodeLP*> group;
        std::vector<char> skip(frontier.size(), 0);
        for (size_t i = 0; i < frontier.size(); ++i) {
            if (failed || c.over_budget()) { skip[i] = 1; c.stop = true; continue; }
            c.budget_used++;
            if (c.count_work) c.out->Nodes++;
            if (frontier[i].depth > MaxDepth) { skip[i] = 2; continue; }
            if (c.opt.test_node_lp) prepare(c, make_node(*c.root, frontier[i].cuts), lps[i]); else prepare_device(c, frontier[i].cuts, lps[i]);
            group.push_back(&lps[i]);
        }
        c.levels++;
        try { solve_group(c, group, c.root->NumVars()); }
        catch (const LpxException& e) {
            // a rank that leaves the loop would leave its peers waiting in the level's all-reduce: keep taking part, tell them
            // (also from the replicated warm-up: the peers meet this rank in their first all-reduce after the hand-out)
            if (!sharded) throw;
            failed = true; fail_msg = e.what(); fail_code = e.code; c.stop = true; replicated = false;
            for (size_t i = 0; i < frontier.size(); ++i) if (!skip[i]) skip[i] = 1;
        }
        std::vector<FNode> next;
        precompute_feas(c, lps, frontier, skip);
        for (size_t i = 0; i < frontier.size(); ++i) {
            if (skip[i] == 1) continue;
            if (skip[i] == 2) { node_log(c, frontier[i].depth, O_DEPTH, -1, 0.0); continue; }
            int fl = 0, ce = 0;
            int k = decide(c, frontier[i].cuts, lps[i], frontier[i].depth, "Node", fl, ce);
            if (k < 0) continue;
            FNode up{frontier[i].cuts, frontier[i].depth + 1}; up.cuts.push_back({k, Rel::GE, (double)ce});
            FNode dn{frontier[i].cuts, frontier[i].depth + 1}; dn.cuts.push_back({k, Rel::LE, (double)fl});
            next.push_back(std::move(up));                                   // ceil child first
            next.push_back(std::move(dn));
        }
        frontier.swap(next);
        for (FNode& f : parked) frontier.push_back(std::move(f));
        if (c.stop) frontier.clear();
        if (!sharded) {
            if (frontier.empty() || c.stop) break;
            continue;
        }
        if (replicated) {
            if (!(frontier.empty() || c.stop)) continue;
            // the search ended inside the replicated warm-up: the same on every rank if all is well -- one all-reduce says so
            agree_h = frontier_fingerprint(frontier, false);
            replicated = false;
        }
        // ---- X1: ONE all-reduce(max) per level: incumbent bound, "someone still has work", "someone failed", the deepest
        //      pooled node, and every rank's pool size (slot r; -1 = this rank takes no more nodes) for the rebalancing below;
        //      in a rank's FIRST one also the fingerprint of its replicated phase as {+h, -h}
        std::vector<double> vals(6 + (size_t)world, -INFINITY);
        int maxd = 0; for (const FNode& f : frontier) maxd = std::max(maxd, f.depth);
        vals[0] = c.has_best ? c.best : -INFINITY;
        vals[1] = frontier.empty() ? 0.0 : 1.0;
        vals[2] = failed ? 1.0 : 0.0;
        vals[3] = (double)maxd;
        // a rank whose share of the budget is used up takes no more nodes either: what it received would be dropped at the next level
        vals[4 + rank] = (c.stop || c.over_budget()) ? -1.0 : (double)frontier.size();
        const bool first = !agree_sent;
        if (first && !failed) { vals[4 + world] = agree_h; vals[5 + world] = -agree_h; }
        agree_sent = true;
        const double mine = vals[0];
        ex.max(vals.data(), (int)vals.size());
        if (first && vals[2] <= 0.0 && vals[4 + world] != -vals[5 + world]) {
            failed = true; fail_code = LPX_EDEVICE; c.stop = true;
            fail_msg = "sharded search: the replicated warm-up ended differently on different ranks (the node LPs must be bit-identical across ranks)";
            break;
        }
        if (vals[0] > mine) { if (!(c.has_best && c.best >= vals[0])) { c.best = vals[0]; c.has_best = true; c.best_x.clear(); c.best_key.clear(); } }
        if (vals[2] > 0.0) { if (!failed) { failed = true; fail_code = LPX_EDEVICE; fail_msg = "sharded search: a peer rank failed"; } break; }
        if (vals[1] == 0.0) break;
        // ---- rebalancing: node descriptors (branching rows) move, tableaux never do -- a node is rebuilt from the root model
        //      wherever it is solved, exactly as the reference re-solves every node from scratch (Models/Branch&Bound.cs:148,233-248)
        std::vector<int64_t> size(world);
        int64_t total = 0, smax = 0, smin = INT64_MAX; int active = 0;
        for (int r = 0; r < world; ++r) { size[r] = (int64_t)vals[4 + r]; if (size[r] >= 0) { total += size[r]; smax = std::max(smax, size[r]); smin = std::min(smin, size[r]); ++active; } }
        const int64_t slack = std::max<int64_t>(2, c.opt.concurrent_nodes);
        if (active >= 2 && ((smin == 0 && smax >= 2) || smax - smin > 2 * slack)) {
            std::vector<int64_t> fin(world, -1);
            { int64_t base = total / active, rem = total % active; int a = 0;
              for (int r = 0; r < world; ++r) if (size[r] >= 0) { fin[r] = base + (a < rem ? 1 : 0); ++a; } }
            struct Move { int from, to; int64_t count; };
            std::vector<Move> moves;
            { int d = 0, t = 0; std::vector<int64_t> give(world, 0), take(world, 0);
              for (int r = 0; r < world; ++r) if (size[r] >= 0) { give[r] = std::max<int64_t>(0, size[r] - fin[r]); take[r] = std::max<int64_t>(0, fin[r] - size[r]); }
              while (d < world && t < world) {
                  if (give[d] == 0) { ++d; continue; }
                  if (take[t] == 0) { ++t; continue; }
                  const int64_t k = std::min(give[d], take[t]);
                  moves.push_back({d, t, k}); give[d] -= k; take[t] -= k;
              } }
            int64_t nmoved = 0; for (const Move& mv : moves) nmoved += mv.count;
            if (nmoved > 0) {
                const int D = (int)vals[3];
                const size_t stride = 1 + 3 * (size_t)std::max(D, 1);
                std::vector<double> buf((size_t)nmoved * stride, -INFINITY);
                size_t slot = 0;
                for (const Move& mv : moves) {
                    if (mv.from == rank)
                        for (int64_t k = 0; k < mv.count; ++k) {                 // donors give from the end of their pool
                            const FNode f = std::move(frontier.back()); frontier.pop_back();
                            double* b = &buf[(slot + (size_t)k) * stride];
                            b[0] = (double)f.cuts.size();
                            for (size_t e = 0; e < f.cuts.size(); ++e) { b[1 + 3 * e] = f.cuts[e].var; b[2 + 3 * e] = (double)(int)f.cuts[e].rel; b[3 + 3 * e] = f.cuts[e].bound; }
                        }
                    slot += (size_t)mv.count;
                }
                ex.max(buf.data(), (int)buf.size());
                slot = 0;
                for (const Move& mv : moves) {
                    if (mv.to == rank)
                        for (int64_t k = 0; k < mv.count; ++k) {
                            const double* b = &buf[(slot + (size_t)k) * stride];
                            FNode f; const int nc = (int)b[0]; f.depth = nc;
                            for (int e = 0; e < nc; ++e) f.cuts.push_back({(int)b[1 + 3 * e], (Rel)(int)b[2 + 3 * e], b[3 + 3 * e]});
                            frontier.push_back(std::move(f));
                        }
                    slot += (size_t)mv.count;
                }
                c.rebalances++; c.moved += nmoved;
            }
        }
    }
    if (sharded && !replicated && !failed) publish_best(c, ex, c.root->NumVars());
    c.out->Aux = {(double)c.levels, (double)c.allreduces, (double)c.rebalances, (double)c.moved};
    if (failed) throw LpxException(fail_code ? fail_code : LPX_EDEVICE, fail_msg);
}

// ---- search 2: level-synchronous frontier with WARM-STARTED children ----------------------
// Not the reference's algorithm (it re-solves every node from the slack basis): a child starts from its
// parent's final tableau plus the branching row written in the parent's basis, which is dual feasible, so only
// the dual loop runs -- typically a handful of pivots instead of hundreds.  Same LP optimum per node, hence the
// same B&B optimum; the optimal VERTEX of a degenerate node may differ from a cold solve, so node order and
// counts are not comparable with searches 0/1.  Sharding and the per-level all-reduce are those of search 1.
struct WNode { std::vector<Cut> cuts; int depth; lpx_store* store; int slot; int prow; };

void WarmSearch(Ctx& c)
{
    const int world = std::max(1, c.opt.world), rank = c.opt.rank;
    const size_t want = (size_t)world * 2;
    bool replicated = world > 1;
    c.count_work = !(replicated && rank != 0);
    c.budget_used = 0; c.budget_cap = c.opt.max_nodes; c.tie_by_key = true;
    Exchange ex{c, world, rank};
    const bool sharded = (world > 1 || c.opt.shard_one) && (bool)c.opt.allreduce_max;
    bool failed = false; std::string fail_msg; int fail_code = 0;
    const bool root_dual = has_ge_or_eq(*c.root);
    if (!ensure_template(c, root_dual)) return;
    const int w = root_dual ? 1 : 0;
    const int nvars = c.root->NumVars();
    // a parked parent serves two children: release its slot when both are built
    std::map<std::pair<lpx_store*, int>, int> refs;
    auto row_of = [](const NodeLP& lp, int var) { for (size_t i = 0; i < lp.basis_out.size(); ++i) if (lp.basis_out[i] == var) return (int)i; return -1; };
    auto add_children = [&](std::vector<WNode>& out, const std::vector<Cut>& cuts, int depth, const NodeLP& lp, int k, int fl, int ce) {
        const int ik = row_of(lp, k);
        if (ik < 0 || lp.kslot < 0) throw LpxException(LPX_EINVAL, "warm start: branching variable is not basic in the parked parent");
        WNode up{cuts, depth + 1, lp.kstore, lp.kslot, ik}; up.cuts.push_back({k, Rel::GE, (double)ce});
        WNode dn{cuts, depth + 1, lp.kstore, lp.kslot, ik}; dn.cuts.push_back({k, Rel::LE, (double)fl});
        out.push_back(std::move(up)); out.push_back(std::move(dn));
        refs[{lp.kstore, lp.kslot}] = 2;
    };
    auto unref = [&](lpx_store* s, int slot) { auto it = refs.find({s, slot}); if (it != refs.end() && --it->second == 0) { lpx_store_release(s, slot); refs.erase(it); } };

    std::vector<WNode> frontier;
    {   // root: cold solve, result parked
        NodeLP lp; prepare_device(c, std::vector<Cut>{}, lp); lp.keep = true;
        c.budget_used++;
        if (c.count_work) c.out->Nodes++;
        std::vector<NodeLP*> g{&lp};
        try {
            solve_group(c, g, nvars);
            int fl = 0, ce = 0;
            const std::vector<Cut> none;
            int k = decide(c, none, lp, 0, "Root Problem", fl, ce);
            if (k < 0) { if (lp.kslot >= 0) lpx_store_release(lp.kstore, lp.kslot); if (!sharded) return; }   // sharded: on to the one all-reduce that says every rank ended here
            else add_children(frontier, none, 0, lp, k, fl, ce);
        } catch (const LpxException& e) {
            // the peers solve the same root and go on to their first level's all-reduce: meet them there (as LevelSearch does)
            if (!sharded) throw;
            failed = true; fail_msg = e.what(); fail_code = e.code; c.stop = true; replicated = false; frontier.clear();
        }
    }
    double agree_h = -1.0; bool agree_sent = !sharded;      // fingerprint of the end of the replicated phase (see frontier_fingerprint), sent once
    if (sharded && !replicated && !failed) agree_h = frontier_fingerprint(frontier, false);
    for (;;) {
        if (replicated && frontier.size() >= want) {
            agree_h = frontier_fingerprint(frontier, true);
            std::vector<WNode> mine;
            for (size_t i = 0; i < frontier.size(); ++i) {
                if ((int)(i % world) == rank) mine.push_back(std::move(frontier[i]));
                else unref(frontier[i].store, frontier[i].slot);
            }
            frontier.swap(mine);
            replicated = false;
            c.count_work = true;
            if (c.opt.max_nodes > 0) {                                 // the rest of the global budget, split evenly
                const int64_t left = std::max<int64_t>(0, c.opt.max_nodes - c.budget_used);
                c.budget_cap = c.budget_used + (left + world - 1) / world;
            }
        }
        std::vector<WNode> parked;
        if (c.opt.bnb_dive && !replicated && frontier.size() > (size_t)std::max(1, c.opt.concurrent_nodes)) {
            std::stable_sort(frontier.begin(), frontier.end(), [](const WNode& a, const WNode& b) { return a.depth > b.depth; });
            const size_t K = (size_t)std::max(1, c.opt.concurrent_nodes);
            parked.assign(std::make_move_iterator(frontier.begin() + K), std::make_move_iterator(frontier.end()));
            frontier.resize(K);
        }
        std::vector<NodeLP> lps(frontier.size());
        std::vector<NodeLP*> group;
        std::vector<char> skip(frontier.size(), 0);
        for (size_t i = 0; i < frontier.size(); ++i) {
            if (failed || c.over_budget()) { skip[i] = 1; c.stop = true; continue; }
            c.budget_used++;
            if (c.count_work) c.out->Nodes++;
            if (frontier[i].depth > MaxDepth) { skip[i] = 2; continue; }
            NodeLP& lp = lps[i];
            const Cut& k = frontier[i].cuts.back();
            lp.warm = true; lp.dual = true; lp.keep = true; lp.wslot = w; lp.depth = frontier[i].depth;
            lp.pstore = frontier[i].store; lp.pslot = frontier[i].slot; lp.prow = frontier[i].prow;
            lp.cvar.push_back(k.var); lp.wis_ge = (k.rel == Rel::GE); lp.wbound = k.bound;
            lp.R = c.tplR[w] + frontier[i].depth; lp.C = c.tplC[w] + frontier[i].depth;
            lp.cuts_for_feas = &frontier[i].cuts;
            group.push_back(&lp);
        }
        std::vector<WNode> next;
        try {
            solve_group(c, group, nvars);
            for (const WNode& f : frontier) unref(f.store, f.slot);      // every child of this level is built
            { const double t0 = PhaseTimer::now(); precompute_feas(c, lps, frontier, skip); g_pt.decide += PhaseTimer::now() - t0; }
            for (size_t i = 0; i < frontier.size(); ++i) {
                if (skip[i] == 1) continue;
                if (skip[i] == 2) { node_log(c, frontier[i].depth, O_DEPTH, -1, 0.0); continue; }
                int fl = 0, ce = 0;
                const double t0 = PhaseTimer::now();
                int k = decide(c, frontier[i].cuts, lps[i], frontier[i].depth, "Node", fl, ce);
                g_pt.decide += PhaseTimer::now() - t0;
                if (k < 0) { if (lps[i].kslot >= 0) lpx_store_release(lps[i].kstore, lps[i].kslot); continue; }
                add_children(next, frontier[i].cuts, frontier[i].depth, lps[i], k, fl, ce);
            }
        } catch (const LpxException& e) {
            // LPX_ENOMEM is the plausible one here (thousands of ~8 MB parent tableaux are parked).  A rank that left now would
            // leave its peers waiting in this level's all-reduce: keep taking part with `failed` up, then stop together.  The
            // parked tableaux go with the stores when the search's context is destroyed.
            if (!sharded) throw;
            failed = true; fail_msg = e.what(); fail_code = e.code; c.stop = true; replicated = false;
            next.clear(); parked.clear();
        }
        frontier.swap(next);
        for (WNode& f : parked) frontier.push_back(std::move(f));
        c.levels++;
        if (replicated && sharded && (frontier.empty() || c.stop)) {     // ended inside the replicated warm-up: one all-reduce says every rank did
            agree_h = frontier_fingerprint(frontier, false);
            replicated = false;
        }
        // X1: incumbent, "someone still has work", "someone failed"; in a rank's first one also the fingerprint {+h, -h} of its replicated phase
        double vals[5] = {c.has_best ? c.best : -INFINITY, (frontier.empty() || c.stop) ? 0.0 : 1.0, failed ? 1.0 : 0.0, -INFINITY, -INFINITY};
        if (!replicated && sharded) {
            const bool first = !agree_sent;
            if (first && !failed) { vals[3] = agree_h; vals[4] = -agree_h; }
            agree_sent = true;
            const double mine = vals[0];
            ex.max(vals, 5);
            if (vals[0] > mine) { if (!(c.has_best && c.best >= vals[0])) { c.best = vals[0]; c.has_best = true; c.best_x.clear(); c.best_key.clear(); } }
            if (vals[2] > 0.0) { if (!failed) { failed = true; fail_code = LPX_EDEVICE; fail_msg = "sharded search: a peer rank failed"; } break; }
            if (first && vals[3] != -vals[4]) {
                failed = true; fail_code = LPX_EDEVICE; c.stop = true;
                fail_msg = "sharded search: the replicated warm-up ended differently on different ranks (the node LPs must be bit-identical across ranks)";
                break;
            }
        }
        if (vals[1] == 0.0) break;
    }
    if (sharded && !replicated && !failed) publish_best(c, ex, nvars);
    c.out->Aux = {(double)c.levels, (double)c.allreduces, 0.0, 0.0};
    if (failed) throw LpxException(fail_code ? fail_code : LPX_EDEVICE, fail_msg);
}

}  // namespace

// BranchAndBound.Solve, Models/Branch&Bound.cs:30-123
SimplexResult BranchAndBound::Solve(const LPProblem& problem, UpdatePivot updatePivot)
{
    BestObjective = -INFINITY; BestSolution.clear(); HasBest = false;
    SimplexResult out;
    struct Whole { double t0 = PhaseTimer::now(); ~Whole() { g_pt.total += PhaseTimer::now() - t0; } } whole;
    Ctx c; c.root = &problem; c.opt = opt; c.cb = updatePivot; c.out = &out;
    c.log("=== Branch & Bound Algorithm ===");
    const char* rootAlgo = has_ge_or_eq(problem) ? "Dual Simplex" : "Primal Simplex";     // :50
    c.log(std::string("Branch & Bound: Using ") + rootAlgo + " for the ROOT LP relaxation.");

    EngineOptions lopt = opt; lopt.dual_flags = opt.bnb_mode == 1 ? LPX_DUAL_REPAIRED : 0;
    lopt.quiet = true;                  // the root LP's Report text is not part of the B&B result (:99-122)
    LPSolver solver(lopt);
    SimplexResult rootRes;
    out.LpSolves = 1;
    try {
        if (opt.test_node_lp) {             // test seam: no device, no root tableau in the result
            NodeLP lp; prepare(c, problem, lp);
            std::vector<NodeLP*> g{&lp};
            out.LpSolves = 0;
            solve_group(c, g, problem.NumVars());
            if (lp.error) throw LpxException(LPX_EINVAL, "root relaxation failed");
            rootRes.HasSolution = lp.has_solution; rootRes.Solution = lp.x; rootRes.OptimalValue = lp.z;
        } else
        { const double t0 = PhaseTimer::now(); rootRes = solver.Solve(problem, rootAlgo, nullptr); g_pt.root += PhaseTimer::now() - t0; }   // :57
    } catch (const LpxException& ex) {
        if (ex.code == LPX_EDEVICE || ex.code == LPX_ENOMEM) throw;
        c.log(std::string("Root Problem: LP relaxation infeasible or error: ") + ex.what());
        out.Report = "LP relaxation infeasible"; out.Summary = "Error: Infeasible"; out.Status = LPX_INFEASIBLE;
        return out;                                                                       // :59-63
    }
    out.Stats.pivots += rootRes.Stats.pivots;
    if (!rootRes.HasSolution) {                                                           // :66-70
        c.log("Root Problem: Invalid Simplex result (missing Solution, Tableau, Basis, or VarNames).");
        out.Report = "Invalid Simplex result"; out.Summary = "Error: Invalid result"; out.Status = LPX_INFEASIBLE;
        return out;
    }
    std::vector<double> xRoot(rootRes.Solution.begin(), rootRes.Solution.begin() + std::min<size_t>(rootRes.Solution.size(), problem.NumVars()));
    const double zRoot = rootRes.OptimalValue;
    c.log("Root Problem LP solution: z* = " + FormatF(zRoot, 3));

    auto BuildReport = [&]() {                                                            // :99-122
        std::string sb = "Branch & Bound Finished.\n";
        if (!c.has_best) sb += "No integer-feasible solution found.\n";
        else {
            sb += "Best integer z* = " + FormatF(c.best, 3) + "\n";
            sb += "Best integer x* = [";
            for (size_t i = 0; i < c.best_x.size(); ++i) { if (i) sb += ", "; sb += FormatF(c.best_x[i], 3); }
            sb += "]\n";
        }
        out.Report = sb; out.Summary = sb;
        out.OptimalValue = c.has_best ? c.best : -INFINITY;
        out.Solution = c.best_x; out.HasSolution = true;
        out.Tableau = rootRes.Tableau; out.R = rootRes.R; out.C = rootRes.C;
        out.Basis = rootRes.Basis; out.VarNames = rootRes.VarNames;
        out.Status = c.has_best ? LPX_OPTIMAL : LPX_INFEASIBLE;
        BestObjective = out.OptimalValue; BestSolution = c.best_x; HasBest = c.has_best;
    };

    if (IsIntegral(xRoot) && IsFeasible(xRoot, problem, {})) {                                // :85-91
        set_incumbent(c, xRoot, zRoot);
        c.log("Root Problem is already integral and feasible. Branch & Bound not required.");
        BuildReport();
        return out;
    }
    c.log("Root solution is fractional -> starting Branch & Bound.");
    if (opt.bnb_search == 0) {
        std::vector<Cut> cuts;
        SolveNode(c, cuts, 0, "Root Problem");                                            // :95 (root solved a second time)
    } else if (opt.bnb_search == 2 && !opt.test_node_lp && opt.bnb_mode == 1) {
        WarmSearch(c);
    } else {
        LevelSearch(c);
    }
    BuildReport();
    return out;
}

}}  // namespace lpx::host
