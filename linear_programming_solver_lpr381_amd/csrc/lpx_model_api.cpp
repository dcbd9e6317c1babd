// lpx_model_api.cpp -- model-level C ABI (lpx_solve, lpx_parse_text) over the C++ host mirror.
#include "lpx_internal.h"
#include "host/model.h"
#include "../../include/lpx_test.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace lpx;
using namespace lpx::host;

namespace lpx { bool comm_active(int* rank, int* world); }

namespace {

char* dup_str(const std::string& s) { char* p = (char*)std::malloc(s.size() + 1); std::memcpy(p, s.c_str(), s.size() + 1); return p; }
template <class T> T* dup_vec(const std::vector<T>& v) {
    T* p = (T*)std::malloc(sizeof(T) * (v.size() ? v.size() : 1));
    if (!v.empty()) std::memcpy(p, v.data(), sizeof(T) * v.size());
    return p;
}

LPProblem to_problem(const lpx_problem* p)
{
    LPProblem q;
    q.ObjectiveSense = p->sense == LPX_MIN ? Sense::Min : Sense::Max;
    q.C.assign(p->c, p->c + p->n);
    for (int i = 0; i < p->m; ++i) {
        Constraint c;
        c.A.assign(p->A + (size_t)i * p->n, p->A + (size_t)(i + 1) * p->n);
        c.Relation = p->rel[i] == LPX_GE ? Rel::GE : (p->rel[i] == LPX_EQ ? Rel::EQ : Rel::LE);
        c.B = p->b[i];
        q.Constraints.push_back(std::move(c));
    }
    return q;
}

}  // namespace

// test-only stand-ins for the device loops (include/lpx_test.h); never set by a product path
static thread_local lpx_test_seams g_seams = {nullptr, nullptr, nullptr, 0};
extern "C" void lpx_test_set_seams(const lpx_test_seams* s)
{
    if (s) g_seams = *s; else g_seams = lpx_test_seams{nullptr, nullptr, nullptr, 0};
}

static EngineOptions to_engine(const lpx_solve_opts* o)
{
    EngineOptions e;
    e.max_iter = o->max_iter > 0 ? o->max_iter : 10000;
    e.batch = o->batch; e.render_iterations = o->render_iterations != 0;
    e.dual_flags = o->dual_flags; e.bnb_mode = o->bnb_mode; e.bnb_search = o->bnb_search;
    e.concurrent_nodes = o->concurrent_nodes > 0 ? o->concurrent_nodes : 1;
    e.rank = o->rank; e.world = o->world > 0 ? o->world : 1; e.max_nodes = o->max_nodes;
    e.bnb_dive = o->bnb_dive;
    if (o->allreduce_max) {
        auto fn = o->allreduce_max; void* u = o->allreduce_user;
        e.allreduce_max = [fn, u](double* v, int n) { fn(u, v, n); };
    } else {
        // no host callback: the library's own RCCL communicator (lpx_comm_init) carries the exchange
        int crank = 0, cworld = 0;
        if (comm_active(&crank, &cworld)) {
            if (e.world != cworld || e.rank != crank)
                throw LpxException(LPX_EINVAL, "lpx_solve: opts.rank / opts.world (" + std::to_string(e.rank) + " / " + std::to_string(e.world) +
                                               ") differ from the communicator's (" + std::to_string(crank) + " / " + std::to_string(cworld) + ")");
            e.allreduce_max = [](double* v, int n) {
                const int rc = lpx_comm_allreduce_max(v, n);
                if (rc) { char b[512]; lpx_last_error(b, sizeof(b)); throw LpxException(rc, std::string("liblpx: ") + b); }
            };
            // LPX_COMM_SHARD_ONE=1 (diagnostic): a world of ONE still runs the sharded code path -- hand-out, one all-reduce per
            // level / round through RCCL, publication of x -- so that path can be exercised on a single GPU
            static const bool shard_one = [] { const char* v = std::getenv("LPX_COMM_SHARD_ONE"); return v && v[0] == '1'; }();
            e.shard_one = shard_one && cworld == 1;
        }
    }
    if (g_seams.node_lp) {
        auto fn = g_seams.node_lp; void* u = g_seams.user;
        e.test_node_lp = [fn, u](double* T, int R, int C, int32_t* basis, int dual, int repaired, int max_iter, int nvars,
                                 double* x, double* z, int64_t* pivots) { return fn(u, T, R, C, basis, dual, repaired, max_iter, nvars, x, z, pivots); };
    }
    e.test_fail_after_nodes = g_seams.fail_after_nodes;
    if (g_seams.knap_relax) {
        auto fn = g_seams.knap_relax; void* u = g_seams.user;
        e.test_knap_relax = [fn, u](int count, const int32_t* off, const int32_t* fidx, const int8_t* fval, double* profit,
                                    double* weight, int32_t* frac, double* fracval) { return fn(u, count, off, fidx, fval, profit, weight, frac, fracval); };
    }
    return e;
}

static UpdatePivot to_callback(const lpx_solve_opts* o)
{
    UpdatePivot cb;
    if (o->text_cb) {
        auto fn = o->text_cb; void* u = o->text_user;
        cb = [fn, u](const std::string& text, const Highlight* h) {
            fn(u, text.c_str(), h ? h->cells.data() : nullptr, h ? h->R : 0, h ? h->C : 0);
        };
    }
    return cb;
}

static void fill_result(lpx_result* out, const SimplexResult& r, int nvars)
{
    out->status = r.Status;
    out->has_solution = r.HasSolution ? 1 : 0;
    out->optimal_value = r.OptimalValue;
    out->n = (int)r.Solution.size(); out->x = dup_vec(r.Solution);
    out->R = r.R; out->C = r.C; out->T = dup_vec(r.Tableau);
    out->basis = dup_vec(r.Basis);
    out->n_pivots = (int)(r.Trace.size() / 2); out->trace = dup_vec(r.Trace);
    out->report = dup_str(r.Report); out->summary = dup_str(r.Summary);
    out->lp_solves = r.LpSolves; out->nodes = r.Nodes;
    out->n_log = (int)(r.NodeLog.size() / 3); out->node_log = dup_vec(r.NodeLog); out->node_z = dup_vec(r.NodeZ);
    for (size_t i = 0; i < 4 && i < r.NodeZ.size() && r.NodeLog.empty(); ++i) out->aux[i] = r.NodeZ[i];
    for (size_t i = 0; i < 4 && i < r.Aux.size(); ++i) out->aux[i] = r.Aux[i];
    out->stats = r.Stats;
    out->n_cuts = nvars >= 0 ? (int)(r.Cuts.size() / (size_t)(nvars + 1)) : 0; out->cuts = dup_vec(r.Cuts);
}

// ---- SensitivityAnalysis over caller-owned arrays (include/lpx.h) -----------------------------------------
namespace {
struct SensCtx {
    LPProblem q; SimplexResult r;
    SensCtx(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis)
    {
        q = to_problem(p);
        r.HasSolution = T != nullptr;
        if (T && R > 0 && C > 0) r.Tableau.assign(T, T + (size_t)R * C);
        r.R = R; r.C = C;
        if (basis && R > 1) r.Basis.assign(basis, basis + (R - 1));
        const int n = p->n, ns = C - 1 - n;                       // VarNames as BuildTableau names them, PrimalSimplex.cs:200-201
        for (int j = 0; j < n; ++j) r.VarNames.push_back("x" + std::to_string(j + 1));
        for (int j = 0; j < ns; ++j) r.VarNames.push_back("c" + std::to_string(j + 1));
    }
};
int copy_out(const std::string& s, char* buf, int len)
{
    if (buf && len > 0) { std::strncpy(buf, s.c_str(), (size_t)len - 1); buf[len - 1] = 0; }
    return (int)s.size();
}
template <class F> int guarded(const char* what, F&& f)
{
    try { return f(); }
    catch (const LpxException& ex) { set_error(ex.what()); return ex.code; }
    catch (const std::exception& ex) { set_error(std::string(what) + ": " + ex.what()); return LPX_EINVAL; }
}
}  // namespace

extern "C" {

void lpx_default_solve_opts(lpx_solve_opts* o)
{
    std::memset(o, 0, sizeof(*o));
    o->max_iter = 10000; o->concurrent_nodes = 1; o->world = 1;
}

int lpx_solve(const lpx_problem* p, const char* algorithm, const lpx_solve_opts* o, lpx_result* out)
{
    if (!p || !algorithm || !out) { set_error("lpx_solve: null argument"); return LPX_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    lpx_solve_opts d; if (!o) { lpx_default_solve_opts(&d); o = &d; }
    try {
        EngineOptions e = to_engine(o);
        UpdatePivot cb = to_callback(o);
        LPProblem q = to_problem(p);
        SimplexResult r;
        // Form1.btnSolve_Click (Form1.cs:249-261) builds the two cutting-plane solvers itself; LPSolver does not
        // know them (Models/LPSolver.cs:18-43), so they are routed here and not in the LPSolver mirror.
        const std::string key = LPSolver::NormalizeAlgorithmKey(algorithm);
        if (key == "cutting plane") r = CuttingPlane(e).Solve(q, cb);
        else if (key == "revised cutting plane" || key == "cutting plane revised") r = CuttingPlaneRevised(e).Solve(q, cb);
        else r = LPSolver(e).Solve(q, algorithm, cb);
        fill_result(out, r, p->n);
        return 0;
    } catch (const LpxException& ex) {
        set_error(ex.what());
        return ex.code;
    } catch (const std::exception& ex) {
        set_error(std::string("lpx_solve: ") + ex.what());
        return LPX_EINVAL;
    }
}

int lpx_sensitivity_range_report(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                 const char* target, char* buf, int len)
{
    if (!p || !target) { set_error("lpx_sensitivity_range_report: null argument"); return LPX_EINVAL; }
    return guarded("lpx_sensitivity_range_report", [&]() -> int {
        SensCtx c(p, T, R, C, basis);
        SensitivityAnalysis sa(&c.q, &c.r);
        return copy_out(sa.GetRangeReport(target), buf, len);
    });
}

int lpx_sensitivity_range(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                          const char* target, double* mn, double* mx)
{
    if (!p || !target || !mn || !mx) { set_error("lpx_sensitivity_range: null argument"); return LPX_EINVAL; }
    return guarded("lpx_sensitivity_range", [&]() -> int {
        SensCtx c(p, T, R, C, basis);
        SensitivityAnalysis sa(&c.q, &c.r);
        std::pair<double, double> r = sa.Range(target);
        *mn = r.first; *mx = r.second;
        return 0;
    });
}

int lpx_sensitivity_apply_change(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                 const char* target, double value, int* field, int* index, char* buf, int len)
{
    if (!p || !target) { set_error("lpx_sensitivity_apply_change: null argument"); return LPX_EINVAL; }
    return guarded("lpx_sensitivity_apply_change", [&]() -> int {
        SensCtx c(p, T, R, C, basis);
        SensitivityAnalysis sa(&c.q, &c.r);
        int f = -1, ix = -1;
        sa.Locate(target, &f, &ix);
        std::string msg = sa.ApplyChange(target, value);
        if (field) *field = f;
        if (index) *index = ix;
        return copy_out(msg, buf, len);
    });
}

int lpx_sensitivity_shadow_prices(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                  char* buf, int len)
{
    if (!p) { set_error("lpx_sensitivity_shadow_prices: null argument"); return LPX_EINVAL; }
    return guarded("lpx_sensitivity_shadow_prices", [&]() -> int {
        SensCtx c(p, T, R, C, basis);
        SensitivityAnalysis sa(&c.q, &c.r);
        return copy_out(sa.GetShadowPricesReport(), buf, len);
    });
}

int lpx_sensitivity_solve_duality(const lpx_problem* p, const double* T, int R, int C, const int32_t* basis,
                                  const lpx_solve_opts* o, lpx_result* out)
{
    if (!p || !out) { set_error("lpx_sensitivity_solve_duality: null argument"); return LPX_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    lpx_solve_opts d; if (!o) { lpx_default_solve_opts(&d); o = &d; }
    return guarded("lpx_sensitivity_solve_duality", [&]() -> int {
        SensCtx c(p, T, R, C, basis);
        SensitivityAnalysis sa(&c.q, &c.r, to_engine(o));
        SimplexResult r = sa.SolveUsingDuality();
        fill_result(out, r, -1);
        return 0;
    });
}

void lpx_result_free(lpx_result* r)
{
    if (!r) return;
    std::free(r->x); std::free(r->T); std::free(r->basis); std::free(r->trace); std::free(r->report);
    std::free(r->summary); std::free(r->node_log); std::free(r->node_z); std::free(r->cuts);
    std::memset(r, 0, sizeof(*r));
}

int lpx_parse_text(const char* text, lpx_parsed* out)
{
    if (!text || !out) { set_error("lpx_parse_text: null argument"); return LPX_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    try {
        LPProblem q = ParseFromText(text);
        const int n = q.NumVars(), m = (int)q.Constraints.size();
        out->sense = q.ObjectiveSense == Sense::Min ? LPX_MIN : LPX_MAX;
        out->n = n; out->m = m;
        out->c = dup_vec(q.C);
        out->A = (double*)std::calloc((size_t)(m ? m : 1) * (n ? n : 1), sizeof(double));
        out->rel = (int32_t*)std::malloc(sizeof(int32_t) * (m ? m : 1));
        out->b = (double*)std::malloc(sizeof(double) * (m ? m : 1));
        for (int i = 0; i < m; ++i) {
            const Constraint& c = q.Constraints[i];
            for (int j = 0; j < n && j < (int)c.A.size(); ++j) out->A[(size_t)i * n + j] = c.A[j];
            if ((int)c.A.size() < n) out->ragged = 1;     // IndexOutOfRange later in BuildTableau (PrimalSimplex.cs:190)
            out->rel[i] = c.Relation == Rel::GE ? LPX_GE : (c.Relation == Rel::EQ ? LPX_EQ : LPX_LE);
            out->b[i] = c.B;
        }
        return 0;
    } catch (const LpxException& ex) {
        set_error(ex.what());
        return ex.code;
    }
}

void lpx_parsed_free(lpx_parsed* p)
{
    if (!p) return;
    std::free(p->c); std::free(p->A); std::free(p->rel); std::free(p->b);
    std::memset(p, 0, sizeof(*p));
}

int lpx_format_number(double v, char* buf, int len)
{
    std::string s = FormatNumber(v);
    if (buf && len > 0) { std::strncpy(buf, s.c_str(), len - 1); buf[len - 1] = 0; }
    return (int)s.size();
}

}  // extern "C"
