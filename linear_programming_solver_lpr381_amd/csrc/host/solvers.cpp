// host/solvers.cpp -- PrimalSimplex / DualSimplex / RevisedPrimalSimplex / LPSolver mirrors.
// Model preparation, tableau construction and report text follow the reference line by line on the
// host; the pivot loops cross the C ABI (include/lpx.h) into the HIP kernels.
#include "model.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

namespace lpx { namespace host {

namespace {
struct ExactHandleCache {
    std::map<std::pair<int, int>, std::vector<lpx_tableau*>> free_;
    size_t count = 0;
    std::mutex mu;
};
ExactHandleCache& exact_cache() { static ExactHandleCache* c = new ExactHandleCache; return *c; }   // never destroyed: the HIP runtime may be gone at exit
constexpr size_t kExactMax = 16;
constexpr size_t kExactBytes = (size_t)32 << 20;
}  // namespace

lpx_tableau* acquire_exact_handle(int R, int C)
{
    {
        ExactHandleCache& c = exact_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        auto it = c.free_.find({R, C});
        if (it != c.free_.end() && !it->second.empty()) { lpx_tableau* t = it->second.back(); it->second.pop_back(); --c.count; return t; }
    }
    lpx_tableau* t = nullptr;
    const int rc = lpx_tableau_create(R, C, &t);
    if (rc) { char buf[1024]; lpx_last_error(buf, sizeof(buf)); throw LpxException(rc, std::string("liblpx: ") + buf); }
    return t;
}

void release_exact_handle(lpx_tableau* t, int R, int C)
{
    if (!t) return;
    static const bool keep = [] { const char* e = std::getenv("LPX_HANDLE_CACHE"); return !(e && e[0] == '0'); }();   // diagnostic
    if (keep && sizeof(double) * (size_t)R * (size_t)C <= kExactBytes) {
        ExactHandleCache& c = exact_cache();
        std::lock_guard<std::mutex> lk(c.mu);
        if (c.count < kExactMax) { c.free_[{R, C}].push_back(t); ++c.count; return; }
    }
    lpx_tableau_destroy(t);
}

namespace {

std::string last_error()
{
    char buf[1024];
    lpx_last_error(buf, sizeof(buf));
    return buf;
}

[[noreturn]] void throw_lib(int rc) { throw LpxException(rc, "liblpx: " + last_error()); }

struct TableauHandle {
    lpx_tableau* h = nullptr; int R_, C_;
    TableauHandle(int R, int C) : R_(R), C_(C) { h = acquire_exact_handle(R, C); }
    ~TableauHandle() { release_exact_handle(h, R_, C_); }
    TableauHandle(const TableauHandle&) = delete;
};

struct CbCtx {
    UpdatePivot cb; bool render; const char* title;
    lpx_tableau* h; int R, C;
    const std::vector<std::string>* varNames;
};

void pivot_event(void* user, int iter, int row, int col)
{
    CbCtx* c = static_cast<CbCtx*>(user);
    if (!c->cb) return;
    if (c->render) {
        std::vector<double> T((size_t)c->R * c->C);
        std::vector<int32_t> basis(c->R - 1);
        if (lpx_tableau_download(c->h, T.data(), basis.data()) != 0) return;
        Highlight hl; hl.R = c->R; hl.C = c->C; hl.cells.assign((size_t)c->R * c->C, 0);
        for (int j = 0; j < c->C; ++j) hl.cells[(size_t)row * c->C + j] = 1;     // pivot row, :118
        for (int i = 0; i < c->R; ++i) hl.cells[(size_t)i * c->C + col] = 1;     // pivot col, :119
        c->cb(AppendTableau(c->title, T.data(), c->R, c->C, basis, *c->varNames, iter), &hl);
    } else {
        c->cb(std::string(c->title) + " " + std::to_string(iter) + ": leaving row " + std::to_string(row) +
              ", entering " + (*c->varNames)[col] + "\n", nullptr);
    }
}

// double.ToString() of .NET Core 3.0+: shortest round-trippable digits
std::string shortest(double v)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "\xE2\x88\x9E" : "-\xE2\x88\x9E";
    if (v == 0) return std::signbit(v) ? "-0" : "0";
    char buf[64];
    for (int p = 1; p <= 17; ++p) {
        std::snprintf(buf, sizeof(buf), "%.*g", p, v);
        if (std::strtod(buf, nullptr) == v) break;
    }
    double a = std::fabs(v);
    if (a >= 1e-5 && a < 1e15) {
        // positional
        std::string s(buf);
        if (s.find('e') != std::string::npos) {
            int digits = 17; std::snprintf(buf, sizeof(buf), "%.*f", digits, v); s = buf;
            // trim to shortest round trip
            while (s.find('.') != std::string::npos && s.back() == '0') s.pop_back();
            if (s.back() == '.') s.pop_back();
        }
        return s;
    }
    std::string s(buf);
    size_t e = s.find('e');
    if (e == std::string::npos) return s;
    std::string mant = s.substr(0, e);
    int ex = std::atoi(s.c_str() + e + 1);
    char eb[16];
    std::snprintf(eb, sizeof(eb), "E%c%02d", ex < 0 ? '-' : '+', std::abs(ex));
    return mant + eb;
}

// FinalizeReport, Models/PrimalSimplex.cs:130-159 / Models/DualSimplex.cs:283-311
void finalize_text(std::string& report, std::string& summary, const std::vector<double>& x, double z, const char* status)
{
    report += std::string("\nStatus: ") + status + "\n";
    for (size_t j = 0; j < x.size(); ++j) report += "  x" + std::to_string(j + 1) + " = " + FormatRound3(x[j]) + "\n";
    report += "  z* = " + FormatRound3(z) + "\n";
    summary += std::string("Status: ") + status + "\n";
    summary += "z* = " + FormatRound3(z) + "\n";
    summary += "x* = [";
    for (size_t j = 0; j < x.size(); ++j) { if (j) summary += ", "; summary += shortest(RoundHalfEven(x[j], 3)); }
    summary += "]\n";
}

const char* status_text(int st)
{
    return st == LPX_OPTIMAL ? "OPTIMAL" : st == LPX_UNBOUNDED ? "UNBOUNDED" : st == LPX_INFEASIBLE ? "INFEASIBLE" : "?";
}

void check_row_lengths(const LPProblem& p)
{
    // row.A[j] for j < n (Models/PrimalSimplex.cs:190): a short row is an IndexOutOfRangeException
    for (const Constraint& c : p.Constraints)
        if ((int)c.A.size() < p.NumVars())
            throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");
}

}  // namespace

// ExpandEqualitiesToInequalities, Models/PrimalSimplex.cs:161-177
LPProblem ExpandEqualitiesToInequalities(const LPProblem& model)
{
    LPProblem expanded;
    expanded.C = model.C;
    for (const Constraint& cons : model.Constraints) {
        if (cons.Relation == Rel::EQ) {
            Constraint pos = cons; pos.Relation = Rel::LE;
            expanded.Constraints.push_back(pos);
            Constraint neg = cons; neg.Relation = Rel::LE;
            for (double& a : neg.A) a *= -1;
            neg.B *= -1;
            expanded.Constraints.push_back(neg);
        } else {
            expanded.Constraints.push_back(cons);
        }
    }
    return expanded;
}

// BuildTableau, Models/PrimalSimplex.cs:179-203 (== Models/DualSimplex.cs:160-189)
void BuildTableauPrimal(const LPProblem& model, std::vector<double>& T, int& R, int& C,
                        std::vector<int32_t>& basis, std::vector<std::string>& varNames)
{
    check_row_lengths(model);
    const int m = (int)model.Constraints.size();
    const int n = model.NumVars();
    const int s = m;
    R = m + 1; C = n + s + 1;
    T.assign((size_t)R * C, 0.0);
    for (int i = 0; i < m; ++i) {
        const Constraint& row = model.Constraints[i];
        for (int j = 0; j < n; ++j) T[(size_t)i * C + j] = row.A[j];
        T[(size_t)i * C + n + i] = 1.0;
        T[(size_t)i * C + n + s] = row.B;
    }
    for (int j = 0; j < n; ++j) T[(size_t)m * C + j] = -model.C[j];
    basis.resize(m);
    for (int i = 0; i < m; ++i) basis[i] = n + i;
    varNames.resize(n + s);
    for (int j = 0; j < n; ++j) varNames[j] = "x" + std::to_string(j + 1);
    for (int j = 0; j < s; ++j) varNames[n + j] = "c" + std::to_string(j + 1);
}

// PrepareForTableau, Models/DualSimplex.cs:117-158.  fix_d1 skips the second sign flip (:148-153).
LPProblem PrepareForTableauDual(const LPProblem& original, bool fix_d1)
{
    const double Eps = 1e-9;
    LPProblem model = original.Clone();
    if (model.ObjectiveSense == Sense::Min) for (double& c : model.C) c = -c;
    LPProblem expanded;
    expanded.C = model.C;
    expanded.ObjectiveSense = Sense::Max;
    for (const Constraint& cons : model.Constraints) {
        if (cons.Relation == Rel::EQ) {
            Constraint pos = cons; pos.Relation = Rel::LE;
            expanded.Constraints.push_back(pos);
            Constraint neg = cons; neg.Relation = Rel::LE; neg.B = -cons.B;
            for (double& a : neg.A) a *= -1;
            expanded.Constraints.push_back(neg);
        } else {
            Constraint row = cons;
            if (row.Relation == Rel::GE) {
                for (double& a : row.A) a *= -1;
                row.B *= -1;
                row.Relation = Rel::LE;
            }
            if (!fix_d1 && row.B < -Eps) {          // defect D1: `x >= b` ends up as `x <= b`
                for (double& a : row.A) a *= -1;
                row.B *= -1;
            }
            expanded.Constraints.push_back(row);
        }
    }
    return expanded;
}

static void fill_solution(SimplexResult& res, std::vector<double>&& T, int R, int C, std::vector<int32_t>&& basis,
                          std::vector<std::string>&& varNames, std::vector<double>& x, double& z)
{
    const int m = R - 1;
    int n = 0;
    for (const std::string& v : varNames) if (!v.empty() && v[0] == 'x') ++n;
    x.assign(n, 0.0);
    for (int i = 0; i < m; ++i) if (basis[i] < n) x[basis[i]] = T[(size_t)i * C + (C - 1)];
    z = T[(size_t)m * C + (C - 1)];
    res.Tableau = std::move(T); res.R = R; res.C = C;
    res.Basis = std::move(basis); res.VarNames = std::move(varNames);
}

static std::vector<int32_t> fetch_trace(lpx_tableau* h)
{
    int n = 0;
    lpx_tableau_trace(h, nullptr, 0, &n);
    std::vector<int32_t> tr(2 * (size_t)std::max(n, 1));
    lpx_tableau_trace(h, tr.data(), n, &n);
    tr.resize(2 * (size_t)n);
    return tr;
}

// ---------------------------------------------------------------------------------------------------
// PrimalSimplex.Solve, Models/PrimalSimplex.cs:57-127
// ---------------------------------------------------------------------------------------------------
SimplexResult PrimalSimplex::Solve(const LPProblem& original, UpdatePivot updatePivot)
{
    LPProblem model = original.Clone();
    if (model.ObjectiveSense == Sense::Min) for (double& c : model.C) c = -c;          // :62-63
    for (const Constraint& cons : model.Constraints) {                                  // :66-77
        if (cons.Relation == Rel::GE)
            throw LpxException(LPX_E_GE_PRESENT, "Constraint contains '>=' sign. The Primal Simplex method cannot handle this. Please try the Dual Simplex algorithm instead.");
        if (cons.B < -1e-9)
            throw LpxException(LPX_E_NEG_RHS, "Constraint has a negative RHS value. The Primal Simplex method cannot handle this. Please try the Dual Simplex algorithm instead.");
    }
    LPProblem tableauModel = ExpandEqualitiesToInequalities(model);                     // :80
    std::string report = opt.quiet ? std::string() : AppendCanonicalForm(tableauModel); // :82
    std::vector<double> T; int R, C; std::vector<int32_t> basis; std::vector<std::string> varNames;
    BuildTableauPrimal(tableauModel, T, R, C, basis, varNames);                         // :85
    if (updatePivot) updatePivot(AppendTableau("TABLEAU Iteration", T.data(), R, C, basis, varNames, 0), nullptr);   // :88-90

    SimplexResult res;
    if (R < 2) {   // no constraints: ChooseEntering finds a negative z entry, ChooseLeaving finds no row
        bool neg = false; for (int j = 0; j < C - 1; ++j) if (T[(size_t)(R - 1) * C + j] < -1e-9) neg = true;
        std::vector<double> x; double z;
        int st = neg ? LPX_UNBOUNDED : LPX_OPTIMAL;
        if (neg) report += "UNBOUNDED\n";
        fill_solution(res, std::move(T), R, C, std::move(basis), std::move(varNames), x, z);
        finalize_text(report, res.Summary, x, z, status_text(st));
        res.Report = report; res.OptimalValue = z; res.Solution = x; res.HasSolution = true; res.Status = st;
        return res;
    }
    TableauHandle th(R, C);
    int rc = lpx_tableau_upload(th.h, T.data(), basis.data());
    if (rc) throw_lib(rc);
    lpx_run_opts o; lpx_default_opts(&o, 0);
    o.max_iter = opt.max_iter;
    o.batch = (updatePivot && opt.render_iterations) ? 1 : opt.batch;
    CbCtx ctx{updatePivot, opt.render_iterations, "TABLEAU Iteration", th.h, R, C, &varNames};
    int st = lpx_primal_run(th.h, &o, updatePivot ? pivot_event : nullptr, &ctx, &res.Stats);
    if (st < 0) throw_lib(st);
    if (st == LPX_ITER_LIMIT) throw LpxException(LPX_ITER_LIMIT, "Iteration limit exceeded.");      // :95-96
    res.Trace = fetch_trace(th.h);
    rc = lpx_tableau_download(th.h, T.data(), basis.data());
    if (rc) throw_lib(rc);
    if (st == LPX_UNBOUNDED) report += "UNBOUNDED\n";                                   // :104
    std::vector<double> x; double z;
    fill_solution(res, std::move(T), R, C, std::move(basis), std::move(varNames), x, z);
    finalize_text(report, res.Summary, x, z, status_text(st));
    res.Report = report; res.OptimalValue = z; res.Solution = x; res.HasSolution = true; res.Status = st;
    return res;
}

// ---------------------------------------------------------------------------------------------------
// DualSimplex.Solve, Models/DualSimplex.cs:15-114
// ---------------------------------------------------------------------------------------------------
SimplexResult DualSimplex::Solve(const LPProblem& original, UpdatePivot updatePivot)
{
    const bool fix_d1 = opt.dual_flags & LPX_DUAL_FIX_D1, fix_d2 = opt.dual_flags & LPX_DUAL_FIX_D2,
               sound = opt.dual_flags & LPX_DUAL_SOUND;
    LPProblem model = PrepareForTableauDual(original, fix_d1);                          // :18
    std::vector<double> T; int R, C; std::vector<int32_t> basis; std::vector<std::string> varNames;
    BuildTableauPrimal(model, T, R, C, basis, varNames);                                // :21
    SimplexResult res;
    std::string report;
    int st = LPX_OPTIMAL;
    if (R >= 2) {
        TableauHandle th(R, C);
        int rc = lpx_tableau_upload(th.h, T.data(), basis.data());
        if (rc) throw_lib(rc);
        lpx_run_opts o; lpx_default_opts(&o, 1);
        o.max_iter = opt.max_iter;
        o.fdf_guard = sound ? opt.max_iter : 100;                                       // :202
        o.cleanup = sound ? 1 : 0;
        o.batch = (updatePivot && opt.render_iterations) ? 1 : opt.batch;
        CbCtx ctx{updatePivot, opt.render_iterations, "DUAL SIMPLEX TABLEAU Iteration", th.h, R, C, &varNames};
        st = lpx_dual_run(th.h, &o, updatePivot ? pivot_event : nullptr, &ctx, &res.Stats);
        if (st < 0) throw_lib(st);
        if (st == LPX_ITER_LIMIT) throw LpxException(LPX_ITER_LIMIT, "Iteration limit exceeded (Dual Simplex).");   // :39
        res.Trace = fetch_trace(th.h);
        rc = lpx_tableau_download(th.h, T.data(), basis.data());
        if (rc) throw_lib(rc);
    }
    if (st == LPX_INFEASIBLE) report += "INFEASIBLE (no entering column found)\n";      // :94
    SimplexResult full;
    std::vector<double> x; double z;
    fill_solution(full, std::move(T), R, C, std::move(basis), std::move(varNames), x, z);
    finalize_text(report, res.Summary, x, z, status_text(st));
    res.Report = report; res.Status = st;
    if (fix_d2) {
        res.OptimalValue = z; res.Solution = x; res.HasSolution = true;
        res.Tableau = std::move(full.Tableau); res.R = R; res.C = C;
        res.Basis = std::move(full.Basis); res.VarNames = std::move(full.VarNames);
    }
    // else: defect D2 -- `new SimplexResult { Report, Summary }` (:310): everything else stays null / 0
    return res;
}

// Per-iteration text of the revised path.  render: the reference's whole BuildIterationBlock (:191-246; m^2
// formatted numbers per iteration, so one iteration per batch and a state download each); otherwise a
// three-line event.  The ratio lines of the reference use the x_B AFTER the pivot (:131,:139) with the d of
// before it -- reproduced as is.
namespace {
struct RevCtx {
    UpdatePivot cb; const std::vector<std::string>* names; bool render; lpx_revised* h; int m, n; double eps;
    std::vector<int32_t> NidxPrev;
};

void revised_block(RevCtx* c, int iter, int row, int col)
{
    const int m = c->m, n = c->n;
    std::vector<int32_t> Bidx(m), Nidx(n); std::vector<double> xB(m), Binv((size_t)m * m); double z = 0;
    if (lpx_revised_result(c->h, Bidx.data(), Nidx.data(), xB.data(), &z) || lpx_revised_binv(c->h, Binv.data())) return;
    if (iter == 0) {
        c->cb(BuildIterationBlock(0, Bidx, Nidx, *c->names, Binv.data(), m, xB, z, nullptr, -1, nullptr, NAN, c->eps), nullptr);
    } else {
        std::vector<double> rc(n + m), d(m), rN(n);
        if (lpx_revised_iteration_view(c->h, rc.data(), d.data())) return;
        for (int j = 0; j < n; ++j) rN[j] = rc[c->NidxPrev[j]];
        Highlight hl; hl.R = m; hl.C = 4; hl.cells.assign((size_t)m * 4, 0);                   // :136-137
        for (int j = 0; j < 4; ++j) hl.cells[(size_t)row * 4 + j] = 1;
        c->cb(BuildIterationBlock(iter, Bidx, Nidx, *c->names, Binv.data(), m, xB, z, &rN, col, &d, xB[row], c->eps), &hl);
    }
    c->NidxPrev = Nidx;
}

void revised_event(void* user, int iter, int row, int col)
{
    RevCtx* c = static_cast<RevCtx*>(user);
    if (!c->cb) return;
    if (c->render) { revised_block(c, iter, row, col); return; }
    c->cb("=== Revised Simplex Iteration " + std::to_string(iter) + " ===\nEntering variable: " +
          (*c->names)[col] + "\nLeaving row: " + std::to_string(row + 1) + "\n\n", nullptr);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// RevisedPrimalSimplex.Solve, Models/RevisedPrimalSimplex.cs:17-145
// ---------------------------------------------------------------------------------------------------
SimplexResult RevisedPrimalSimplex::Solve(const LPProblem& original, UpdatePivot updatePivot)
{
    for (const Constraint& c : original.Constraints)                                    // :19-21
        if (!(c.Relation == Rel::LE && c.B >= -1e-9))
            throw LpxException(LPX_E_REVISED_PRECOND, "Revised Primal Simplex currently supports only <= constraints with RHS >= 0. Use Dual Simplex for models with >= or =.");
    check_row_lengths(original);
    // Standardize (:148-186): Max -> negate C (the loop minimises); the other branches are dead
    // behind the precondition above.
    const int m = (int)original.Constraints.size(), n = original.NumVars();
    std::vector<double> c(n), A((size_t)std::max(m, 1) * std::max(n, 1)), b(std::max(m, 1));
    for (int j = 0; j < n; ++j) c[j] = original.ObjectiveSense == Sense::Max ? -original.C[j] : original.C[j];
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) A[(size_t)i * n + j] = original.Constraints[i].A[j];
        b[i] = original.Constraints[i].B;
    }
    std::vector<std::string> names(n + m);
    for (int j = 0; j < n; ++j) names[j] = "x" + std::to_string(j + 1);
    for (int j = 0; j < m; ++j) names[n + j] = "c" + std::to_string(j + 1);
    if (m < 1 || n < 1) throw LpxException(LPX_EINVAL, "Revised Primal Simplex needs at least one variable and one constraint.");

    lpx_revised* h = nullptr;
    int rc = lpx_revised_create(m, n, A.data(), c.data(), b.data(), &h);
    if (rc) throw_lib(rc);
    struct Guard { lpx_revised* h; ~Guard() { lpx_revised_destroy(h); } } guard{h};
    SimplexResult res;
    lpx_run_opts o; lpx_default_opts(&o, 1);
    o.max_iter = opt.max_iter; o.batch = opt.batch;
    RevCtx rctx{updatePivot, &names, updatePivot && opt.render_iterations, h, m, n, o.eps, {}};
    if (rctx.render) { o.batch = 1; revised_block(&rctx, 0, -1, -1); }                       // :62
    auto ev = &revised_event;
    int st = lpx_revised_run(h, &o, updatePivot ? ev : nullptr, &rctx, &res.Stats);
    if (st < 0) throw_lib(st);
    if (st == LPX_ITER_LIMIT) throw LpxException(LPX_ITER_LIMIT, "Iteration limit exceeded in Revised Primal Simplex.");   // :144
    std::vector<int32_t> Bidx(m), Nidx(n); std::vector<double> xB(m); double zint = 0;
    rc = lpx_revised_result(h, Bidx.data(), Nidx.data(), xB.data(), &zint);
    if (rc) throw_lib(rc);
    { int k = 0; lpx_revised_trace(h, nullptr, 0, &k); res.Trace.resize(2 * (size_t)std::max(k, 1)); lpx_revised_trace(h, res.Trace.data(), k, &k); res.Trace.resize(2 * (size_t)k); }
    // BuildFinalSummary, :264-295
    std::vector<double> x(n, 0.0);
    for (int i = 0; i < m; ++i) if (Bidx[i] < n) x[Bidx[i]] = xB[i];
    double zOriginal = 0;
    for (int j = 0; j < n; ++j) zOriginal += original.C[j] * x[j];                      // :287-289
    const char* status = status_text(st);
    std::string sb = std::string("\nStatus: ") + status + "\n";
    for (int j = 0; j < n; ++j) sb += "  x" + std::to_string(j + 1) + " = " + FormatRound3(x[j]) + "\n";
    std::string summary = std::string("Status: ") + status + "\n";
    summary += "x* = [";
    for (int j = 0; j < n; ++j) { if (j) summary += ", "; summary += shortest(RoundHalfEven(x[j], 3)); }
    summary += "]\n";
    sb += "  z* = " + FormatRound3(zOriginal) + "\n";
    summary += "z* = " + FormatRound3(zOriginal) + "\n";
    res.Report = sb; res.Summary = summary; res.Status = st;
    // The reference returns text only (:294): Solution/Tableau/Basis/VarNames stay null, OptimalValue 0.
    // Engine extras for callers that want numbers without re-parsing the text:
    res.NodeZ = {zOriginal, zint};
    res.Tableau = x;            // x* in R=1 x C=n form, flagged by HasSolution == false
    res.R = 1; res.C = n;
    res.Basis = Bidx;
    return res;
}

std::string FormatShortest(double v) { return shortest(v); }

// ---------------------------------------------------------------------------------------------------
// LPSolver, Models/LPSolver.cs:16-76
// ---------------------------------------------------------------------------------------------------
std::string LPSolver::NormalizeAlgorithmKey(const std::string& algorithm)
{
    bool blank = true;
    for (char ch : algorithm) if (!std::isspace((unsigned char)ch)) blank = false;
    if (blank) throw LpxException(LPX_E_UNKNOWN_ALGO, "No algorithm selected.");
    std::string key;
    for (char ch : algorithm) key += (char)std::tolower((unsigned char)ch);
    for (size_t p; (p = key.find("algorithm")) != std::string::npos;) key.erase(p, 9);
    std::string out; bool sp = false;
    for (char ch : key) {
        if (std::isspace((unsigned char)ch)) { sp = true; continue; }
        if (sp && !out.empty()) out += ' ';
        sp = false; out += ch;
    }
    return out;
}

SimplexResult LPSolver::Solve(const LPProblem& problem, const std::string& algorithm, UpdatePivot updatePivot)
{
    const std::string key = NormalizeAlgorithmKey(algorithm);
    std::unique_ptr<ILPAlgorithm> algo;
    if (key == "primal simplex" || key == "primal") algo.reset(new PrimalSimplex(opt));
    else if (key == "revised primal simplex" || key == "revised primal") algo.reset(new RevisedPrimalSimplex(opt));
    else if (key == "dual simplex" || key == "dual") algo.reset(new DualSimplex(opt));
    else if (key == "branch and bound simplex" || key == "branch and bound" || key == "bnb") algo.reset(new BranchAndBound(opt));
    // not reachable through the reference's LPSolver (it is never instantiated there);
    // offered under its menu name so the knapsack path has an entry point
    else if (key == "branch and bound knapsack" || key == "knapsack") algo.reset(new BranchAndBoundKnapsack(opt));
    // likewise never instantiated by the reference (Form1.cs:263-270 routes its menu entry to BranchAndBound)
    else if (key == "revised branch and bound" || key == "branch and bound revised") algo.reset(new BranchAndBoundRevised(opt));
    else throw LpxException(LPX_E_UNKNOWN_ALGO, "Algorithm not supported: '" + algorithm + "'. Try one of: Primal Simplex, Revised Primal Simplex, Dual Simplex, Branch and Bound Simplex.");
    SimplexResult result = algo->Solve(problem, updatePivot);
    if (result.HasSolution && !result.Tableau.empty()) { FinalTableau = result.Tableau; FinalR = result.R; FinalC = result.C; HasFinalTableau = true; }
    else { FinalTableau.clear(); HasFinalTableau = false; }
    return result;
}

}}  // namespace lpx::host
