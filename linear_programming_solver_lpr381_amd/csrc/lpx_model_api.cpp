// lpx_model_api.cpp -- model-level C ABI (lpx_solve, lpx_parse_text) over the C++ host mirror.
#include "lpx_internal.h"
#include "host/model.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

using namespace lpx;
using namespace lpx::host;

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
    }
    if (o->test_node_lp) {
        auto fn = o->test_node_lp; void* u = o->test_user;
        e.test_node_lp = [fn, u](double* T, int R, int C, int32_t* basis, int dual, int repaired, int max_iter, int nvars,
                                 double* x, double* z, int64_t* pivots) { return fn(u, T, R, C, basis, dual, repaired, max_iter, nvars, x, z, pivots); };
    }
    if (o->test_knap_relax) {
        auto fn = o->test_knap_relax; void* u = o->test_user;
        e.test_knap_relax = [fn, u](int count, const int32_t* off, const int32_t* fidx, const int8_t* fval, double* profit,
                                    double* weight, int32_t* frac, double* fracval) { return fn(u, count, off, fidx, fval, profit, weight, frac, fracval); };
    }
    UpdatePivot cb;
    if (o->text_cb) {
        auto fn = o->text_cb; void* u = o->text_user;
        cb = [fn, u](const std::string& text, const Highlight* h) {
            fn(u, text.c_str(), h ? h->cells.data() : nullptr, h ? h->R : 0, h ? h->C : 0);
        };
    }
    try {
        LPProblem q = to_problem(p);
        LPSolver solver(e);
        SimplexResult r = solver.Solve(q, algorithm, cb);
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
        out->stats = r.Stats;
        return 0;
    } catch (const LpxException& ex) {
        set_error(ex.what());
        return ex.code;
    } catch (const std::exception& ex) {
        set_error(std::string("lpx_solve: ") + ex.what());
        return LPX_EINVAL;
    }
}

void lpx_result_free(lpx_result* r)
{
    if (!r) return;
    std::free(r->x); std::free(r->T); std::free(r->basis); std::free(r->trace); std::free(r->report);
    std::free(r->summary); std::free(r->node_log); std::free(r->node_z);
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
