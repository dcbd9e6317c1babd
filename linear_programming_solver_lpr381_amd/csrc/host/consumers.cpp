// host/consumers.cpp -- the reference's consumers of SimplexResult.Tableau / Basis / VarNames:
// CuttingPlane (Models/CuttingPlane.cs), CuttingPlaneRevised (Models/CuttingPlaneRevised.cs) and
// SensitivityAnalysis (Models/SensitivityAnalysis.cs).  Host logic only: every LP they solve goes through the
// PrimalSimplex / RevisedPrimalSimplex / DualSimplex mirrors and so through the HIP pivot loops; what is checked
// here is that the tableau the library hands back has the reference's layout.
//
// Followed literally, defects included: both CuttingPlane (:109 "row = i + 1 because row 0 is objective") and
// SensitivityAnalysis (:122, :237, :263, :266, :279) index the tableau as if the objective row came first, while
// BuildTableau puts it last (Models/PrimalSimplex.cs:197-199); the Gomory cut is added as sum f_j x_j <= f_0.
#include "model.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdlib>

namespace lpx { namespace host {

namespace {

const char* rel_name(Rel r) { return r == Rel::LE ? "LE" : (r == Rel::GE ? "GE" : "EQ"); }    // enum ToString()

std::string join(const std::vector<std::string>& v, const char* sep)
{
    std::string o;
    for (size_t i = 0; i < v.size(); ++i) { if (i) o += sep; o += v[i]; }
    return o;
}

// string.Join(" + ", a.Select((a, j) => a != 0 ? $"{a:F3}x{j+1}" : null).Where(s => s != null))
std::string nonzero_terms(const std::vector<double>& a, const char* var)
{
    std::vector<std::string> t;
    for (size_t j = 0; j < a.size(); ++j)
        if (a[j] != 0) t.push_back(FormatF(a[j], 3) + var + std::to_string(j + 1));
    return join(t, " + ");
}

int find_fractional(const std::vector<double>& x)          // CuttingPlane.cs:76-89, CuttingPlaneRevised.cs:80-88
{
    for (size_t i = 0; i < x.size(); ++i) {
        const double frac = x[i] - std::floor(x[i]);
        if (frac > 1e-9 && frac < 1 - 1e-9) return (int)i;
    }
    return -1;
}

bool contains_ci(const std::string& hay, const std::string& needle)      // OrdinalIgnoreCase Contains
{
    auto it = std::search(hay.begin(), hay.end(), needle.begin(), needle.end(),
                          [](char a, char b) { return std::tolower((unsigned char)a) == std::tolower((unsigned char)b); });
    return it != hay.end();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// CuttingPlane.Solve, Models/CuttingPlane.cs:13-139
// ---------------------------------------------------------------------------------------------------
SimplexResult CuttingPlane::Solve(const LPProblem& problem, UpdatePivot updatePivot)
{
    PrimalSimplex simplex(opt);
    LPProblem model = problem.Clone();
    std::string report;
    int iteration = 1;
    const int maxIterations = 50;
    SimplexResult out;
    out.Status = LPX_CUT_INCOMPLETE;

    report += "=== Gomory Cutting Plane Algorithm ===\n";                                   // :22-31
    {
        std::vector<std::string> t;
        for (size_t i = 0; i < problem.C.size(); ++i) t.push_back(FormatF(problem.C[i], 3) + "x" + std::to_string(i + 1));
        report += "Objective: Maximize " + join(t, " + ") + "\n";
    }
    report += "Subject to:\n";
    for (const Constraint& c : problem.Constraints)
        report += nonzero_terms(c.A, "x") + " " + rel_name(c.Relation) + " " + FormatF(c.B, 3) + "\n";
    report += "x_j >= 0, integer\n";

    while (iteration <= maxIterations) {                                                     // :32
        report += "\n--- Iteration " + std::to_string(iteration) + " ---\n";
        SimplexResult lp;
        try {
            lp = simplex.Solve(model, updatePivot);                                          // :40
        } catch (const LpxException& ex) {                                                   // :42-50
            if (ex.code == LPX_EDEVICE || ex.code == LPX_ENOMEM) throw;                      // not the reference's exceptions
            report += std::string("Error in PrimalSimplex: ") + ex.what() + "\n";
            out.Report = report; out.Summary = std::string("Error: ") + ex.what();
            out.Status = LPX_CUT_ERROR;
            return out;
        }
        out.LpSolves++;
        out.Stats.pivots += lp.Stats.pivots; out.Stats.loop_ms += lp.Stats.loop_ms;
        report += lp.Report + "\n";                                                          // :51
        if (!lp.HasSolution) {                                                               // :54-62
            report += "Error: Invalid Simplex result.\n";
            out.Report = report; out.Summary = "Error: Invalid Simplex result";
            out.Status = LPX_CUT_ERROR;
            return out;
        }
        std::vector<double> solution(lp.Solution.begin(),                                    // :65
                                     lp.Solution.begin() + std::min<size_t>(lp.Solution.size(), (size_t)problem.NumVars()));
        if ((int)solution.size() != problem.NumVars()) {                                     // :66-74
            report += "Error: Solution length (" + std::to_string(solution.size()) + ") does not match NumVars (" +
                      std::to_string(problem.NumVars()) + ").\n";
            out.Report = report; out.Summary = "Error: Invalid solution length";
            out.Status = LPX_CUT_ERROR;
            return out;
        }
        {
            std::vector<std::string> t;
            for (double x : solution) t.push_back(FormatF(x, 3));
            report += "Current solution: x* = [" + join(t, ", ") + "], z* = " + FormatF(lp.OptimalValue, 3) + "\n";   // :75
        }
        const int fracIndex = find_fractional(solution);                                     // :78-89
        if (fracIndex == -1) {                                                               // :91-104
            report += "All variables integer. Optimal integer solution found.\n";
            std::vector<std::string> t;
            for (double x : solution) t.push_back(FormatF(x, 2));
            out.Report = report;
            out.Summary = "Status: OPTIMAL INTEGER\nz* = " + FormatF(lp.OptimalValue, 2) + "\nx* = [" + join(t, ", ") + "]";
            out.OptimalValue = lp.OptimalValue; out.Solution = solution; out.HasSolution = true;
            out.Tableau = std::move(lp.Tableau); out.R = lp.R; out.C = lp.C;
            out.Basis = std::move(lp.Basis); out.VarNames = std::move(lp.VarNames);
            out.Status = LPX_CUT_INTEGER;
            return out;
        }
        int row = -1;                                                                        // :107-115
        for (size_t i = 0; i < lp.Basis.size(); ++i)
            if (lp.Basis[i] == fracIndex) { row = (int)i + 1; break; }
        if (row == -1) {                                                                     // :116-124
            report += "Error: Variable x" + std::to_string(fracIndex + 1) + " is not basic.\n";
            out.Report = report; out.Summary = "Error: Non-basic fractional variable";
            out.Status = LPX_CUT_ERROR;
            return out;
        }
        // GenerateGomoryCut, :141-162
        Constraint cut;
        cut.A.assign(problem.NumVars(), 0.0);
        cut.Relation = Rel::LE;
        const double* trow = lp.Tableau.data() + (size_t)row * lp.C;
        const double rhs = trow[lp.C - 1];
        const double f0 = rhs - std::floor(rhs);
        for (int j = 0; j < problem.NumVars(); ++j) {
            const double aij = trow[j];
            const double fj = aij - std::floor(aij);
            if (fj > 1e-9) cut.A[j] = fj;
        }
        cut.B = f0;
        report += "Added Gomory cut: " + nonzero_terms(cut.A, "x") + " <= " + FormatF(cut.B, 3) + "\n";   // :129
        out.Cuts.insert(out.Cuts.end(), cut.A.begin(), cut.A.end());                        // engine extra: the cuts
        out.Cuts.push_back(cut.B);
        model.Constraints.push_back(std::move(cut));                                         // :128
        out.Nodes++;
        iteration++;
    }
    report += "Iteration limit reached. Stopping.\n";                                       // :132-137
    out.Report = report; out.Summary = "Status: INCOMPLETE";
    return out;
}

// ---------------------------------------------------------------------------------------------------
// CuttingPlaneRevised.Solve, Models/CuttingPlaneRevised.cs:14-110
// ---------------------------------------------------------------------------------------------------
static bool extract_solution(const std::string& summary, int nVars, std::vector<double>& out)   // :90-110
{
    size_t pos = 0;
    std::string line; bool found = false;
    while (pos <= summary.size()) {
        size_t e = summary.find('\n', pos);
        std::string l = summary.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
        size_t k = 0; while (k < l.size() && std::isspace((unsigned char)l[k])) ++k;
        if (l.compare(k, 6, "x* = [") == 0) { line = l; found = true; break; }
        if (e == std::string::npos) break;
        pos = e + 1;
    }
    if (!found) return false;
    const size_t s = line.find('['), e = line.find(']');
    if (s == std::string::npos || e == std::string::npos || e <= s) return false;
    out.clear();
    std::string body = line.substr(s + 1, e - s - 1);
    size_t p = 0;
    for (;;) {
        size_t c = body.find(',', p);
        std::string tok = body.substr(p, c == std::string::npos ? std::string::npos : c - p);
        char* end = nullptr;
        const double v = std::strtod(tok.c_str(), &end);
        if (end == tok.c_str()) throw LpxException(LPX_E_PARSE, "Input string was not in a correct format.");   // double.Parse
        out.push_back(v);
        if (c == std::string::npos) break;
        p = c + 1;
    }
    out.resize((size_t)nVars, 0.0);                                                          // :106-108
    return true;
}

SimplexResult CuttingPlaneRevised::Solve(const LPProblem& problem, UpdatePivot updatePivot)
{
    RevisedPrimalSimplex solver(opt);
    LPProblem model = problem.Clone();
    std::string report;
    int iter = 1;
    const int MaxIterations = 50;
    SimplexResult out;
    for (;;) {                                                                               // :21
        SimplexResult lp = solver.Solve(model, updatePivot);                                 // :23 (exceptions propagate)
        out.LpSolves++;
        out.Stats.pivots += lp.Stats.pivots; out.Stats.loop_ms += lp.Stats.loop_ms;
        report += "--- Cutting-Plane Iteration " + std::to_string(iter) + " ---\n";
        report += lp.Report + "\n";
        if (!contains_ci(lp.Summary, "Status: OPTIMAL")) {                                   // :27-35
            report += "Stopping: LP not OPTIMAL; cannot continue cutting.\n";
            out.Report = report; out.Summary = "Terminated: LP not OPTIMAL; cutting-plane stopped.";
            out.Status = LPX_CUT_NOT_OPTIMAL;
            return out;
        }
        std::vector<double> x;
        if (!extract_solution(lp.Summary, problem.NumVars(), x)) {                           // :37-46
            report += "Stopping: Could not parse primal solution.\n";
            out.Report = report; out.Summary = "Terminated: could not parse solution.";
            out.Status = LPX_CUT_ERROR;
            return out;
        }
        out.Tableau = x; out.R = 1; out.C = (int)x.size();       // engine extra (the reference returns text only)
        const int fracIndex = find_fractional(x);                                            // :48
        if (fracIndex == -1) {                                                               // :49-57
            report += "All decision variables are integer. Optimal integer solution found.\n";
            out.Report = report;
            out.Summary = lp.Summary;
            for (size_t p; (p = out.Summary.find("Status: OPTIMAL")) != std::string::npos;) {
                out.Summary.replace(p, 15, "Status: OPTIMAL INTEGER");
                break;      // the status line occurs once; Replace would rewrite every occurrence
            }
            out.Status = LPX_CUT_INTEGER;
            return out;
        }
        const double floorVal = std::floor(x[fracIndex] + 1e-12);                            // :59
        Constraint cut;
        cut.A.assign(model.NumVars(), 0.0);
        cut.A[fracIndex] = 1.0;
        cut.Relation = Rel::LE;
        cut.B = floorVal;
        out.Cuts.insert(out.Cuts.end(), cut.A.begin(), cut.A.end());
        out.Cuts.push_back(cut.B);
        model.Constraints.push_back(std::move(cut));                                         // :66
        out.Nodes++;
        report += "Added cut: x" + std::to_string(fracIndex + 1) + " \xE2\x89\xA4 " + FormatShortest(floorVal) +
                  " (current x" + std::to_string(fracIndex + 1) + " = " + FormatNumber(x[fracIndex]) + ")\n";   // :67
        iter++;
        if (iter > MaxIterations) {                                                          // :70-77
            report += "Iteration limit reached.\n";
            out.Report = report; out.Summary = "Iteration limit reached (solution may still be fractional).";
            out.Status = LPX_CUT_INCOMPLETE;
            return out;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// SensitivityAnalysis, Models/SensitivityAnalysis.cs:11-297
// ---------------------------------------------------------------------------------------------------
SensitivityAnalysis::SensitivityAnalysis(LPProblem* problem_, const SimplexResult* result_, const EngineOptions& o)
    : problem(problem_), result(result_), opt(o)
{
    if (!problem) throw LpxException(LPX_EINVAL, "Value cannot be null. (Parameter 'problem')");        // :20
    if (!result) throw LpxException(LPX_EINVAL, "Value cannot be null. (Parameter 'result')");          // :21
    if (!result->HasSolution || result->Tableau.empty()) throw LpxException(LPX_EINVAL, "SimplexResult.Tableau cannot be null.");   // :24-25
    if (result->Basis.empty() && result->R > 1) throw LpxException(LPX_EINVAL, "SimplexResult.Basis cannot be null.");
    if (result->VarNames.empty()) throw LpxException(LPX_EINVAL, "SimplexResult.VarNames cannot be null.");
    const int m = (int)problem->Constraints.size();
    const int n = problem->NumVars() + m + 1;                                                // :37
    if (result->R != m + 1 || result->C != n)                                                // :38-39
        throw LpxException(LPX_EINVAL, "Tableau dimensions invalid. Expected " + std::to_string(m + 1) + " rows, " +
                           std::to_string(n) + " columns, got " + std::to_string(result->R) + " rows, " +
                           std::to_string(result->C) + " columns.");
    if ((int)result->Basis.size() != m)                                                      // :40-41
        throw LpxException(LPX_EINVAL, "Basis length invalid. Expected " + std::to_string(m) + ", got " +
                           std::to_string(result->Basis.size()) + ".");
    if ((int)result->VarNames.size() < problem->NumVars() + m)                               // :42-43
        throw LpxException(LPX_EINVAL, "VarNames length invalid. Expected at least " + std::to_string(problem->NumVars() + m) +
                           ", got " + std::to_string(result->VarNames.size()) + ".");
}

bool SensitivityAnalysis::basis_contains(int col) const
{
    return std::find(result->Basis.begin(), result->Basis.end(), col) != result->Basis.end();
}

int SensitivityAnalysis::constraint_index(const std::string& target) const                  // :54-55 / :84-85
{
    // target.Split(' ')[1] parsed by int.TryParse; "Constraint" alone would throw IndexOutOfRange in the reference
    const size_t sp = target.find(' ');
    const int m = (int)problem->Constraints.size();
    bool ok = sp != std::string::npos;
    long v = 0;
    if (ok) {
        std::string tok = target.substr(sp + 1);
        const size_t sp2 = tok.find(' ');
        if (sp2 != std::string::npos) tok = tok.substr(0, sp2);
        char* end = nullptr;
        v = std::strtol(tok.c_str(), &end, 10);
        ok = !tok.empty() && end && *end == 0;
    }
    if (!ok || v < 1 || v > m)
        throw LpxException(LPX_EINVAL, "Invalid constraint index in '" + target + "'. Expected 1 to " + std::to_string(m) + ".");
    return (int)v - 1;
}

int SensitivityAnalysis::var_column(const std::string& target) const                        // Array.IndexOf(varNames, target)
{
    for (size_t j = 0; j < result->VarNames.size(); ++j) if (result->VarNames[j] == target) return (int)j;
    throw LpxException(LPX_EINVAL, "Variable '" + target + "' not found in VarNames.");
}

static bool blank(const std::string& s)
{
    for (char c : s) if (!std::isspace((unsigned char)c)) return false;
    return true;
}

std::pair<double, double> SensitivityAnalysis::GetConstraintRange(int index) const          // :277-298
{
    const double* T = result->Tableau.data(); const int C = result->C;
    const int row = index + 1;                                                               // :279
    const int n = C - 1;
    const double currentB = T[(size_t)row * C + n];
    double mn = -INFINITY, mx = INFINITY;
    for (int j = 0; j < problem->NumVars(); ++j) {
        if (basis_contains(j)) continue;
        const double aij = T[(size_t)row * C + j];
        if (std::fabs(aij) < 1e-9) continue;
        const double delta = -T[(size_t)row * C + n] / aij;
        if (aij > 0) mx = std::fmin(mx, currentB + delta);
        else mn = std::fmax(mn, currentB + delta);
    }
    return {mn, mx};
}

std::pair<double, double> SensitivityAnalysis::GetBasicVariableObjectiveRange(int basicVarRow) const   // :250-275
{
    const double* T = result->Tableau.data(); const int C = result->C;
    const int col = result->Basis[basicVarRow];
    const int n = C - 1;
    if (col >= problem->NumVars()) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");   // problem.C[col], :256
    const double current = problem->C[col];
    double mn = -INFINITY, mx = INFINITY;
    for (int j = 0; j < n; ++j) {
        if (basis_contains(j)) continue;
        const double aij = T[(size_t)(basicVarRow + 1) * C + j];                             // :263
        if (std::fabs(aij) < 1e-9) continue;
        const double reducedCost = T[j];                                                     // tableau[0, j], :266
        const double delta = -reducedCost / aij;
        if (aij > 0) mx = std::fmin(mx, current + delta);
        else mn = std::fmax(mn, current + delta);
    }
    return {mn, mx};
}

std::pair<double, double> SensitivityAnalysis::GetNonBasicVariableRange(const std::string& varName) const   // :229-248
{
    const int col = var_column(varName);
    if (col >= problem->NumVars()) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");   // :236
    const double current = problem->C[col];
    const double reducedCost = result->Tableau[col];                                         // tableau[0, col], :237
    double mn = -INFINITY, mx = INFINITY;
    if (reducedCost > 0) mx = current + reducedCost;
    else if (reducedCost < 0) mn = current + reducedCost;
    return {mn, mx};
}

std::string SensitivityAnalysis::GetRangeReport(const std::string& target) const            // :47-76
{
    if (blank(target)) throw LpxException(LPX_EINVAL, "Target cannot be empty.");
    const char* le = " \xE2\x89\xA4 ";
    if (target.compare(0, 10, "Constraint") == 0) {
        const int index = constraint_index(target);
        auto r = GetConstraintRange(index);
        return target + ": " + FormatF(r.first, 3) + le + "B" + le + FormatF(r.second, 3);
    }
    const int col = var_column(target);
    if (basis_contains(col)) {
        int row = 0; while (result->Basis[row] != col) ++row;
        auto r = GetBasicVariableObjectiveRange(row);
        return target + " (Basic): " + FormatF(r.first, 3) + le + "c" + le + FormatF(r.second, 3);
    }
    auto r = GetNonBasicVariableRange(target);
    return target + " (Non-Basic): " + FormatF(r.first, 3) + le + "c" + le + FormatF(r.second, 3);
}

// Numeric face of GetRangeReport (same dispatch, :52-75) for callers that want numbers.
std::pair<double, double> SensitivityAnalysis::Range(const std::string& target) const
{
    if (blank(target)) throw LpxException(LPX_EINVAL, "Target cannot be empty.");
    if (target.compare(0, 10, "Constraint") == 0) return GetConstraintRange(constraint_index(target));
    const int col = var_column(target);
    if (basis_contains(col)) {
        int row = 0; while (result->Basis[row] != col) ++row;
        return GetBasicVariableObjectiveRange(row);
    }
    return GetNonBasicVariableRange(target);
}

// Which model entry ApplyChange assigns (:86, :98, :103): field 0 = Constraints[index].B, 1 = C[index].
void SensitivityAnalysis::Locate(const std::string& target, int* field, int* index) const
{
    if (blank(target)) throw LpxException(LPX_EINVAL, "Target cannot be empty.");
    if (target.compare(0, 10, "Constraint") == 0) { *field = 0; *index = constraint_index(target); return; }
    const int col = var_column(target);
    if (col >= problem->NumVars()) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");
    *field = 1; *index = col;
}

std::string SensitivityAnalysis::ApplyChange(const std::string& target, double value)       // :78-107
{
    if (blank(target)) throw LpxException(LPX_EINVAL, "Target cannot be empty.");
    if (target.compare(0, 10, "Constraint") == 0) {
        const int index = constraint_index(target);
        problem->Constraints[index].B = value;
        return "Constraint " + std::to_string(index + 1) + " B-value updated to " + FormatF(value, 3);
    }
    const int col = var_column(target);
    if (col >= problem->NumVars()) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");   // problem.C[col], :98/:103
    const bool basic = basis_contains(col);
    problem->C[col] = value;
    return std::string(basic ? "Basic" : "Non-basic") + " variable " + target + " objective coefficient updated to " + FormatF(value, 3);
}

std::string SensitivityAnalysis::GetShadowPricesReport() const                               // :109-128
{
    const int m = (int)problem->Constraints.size(), nVars = problem->NumVars();
    if (result->C < nVars + m) throw LpxException(LPX_EINVAL, "Tableau does not contain expected slack columns.");
    std::string sb = "Shadow Prices:\n";
    for (int i = 0; i < m; ++i) {
        const double shadow = -result->Tableau[nVars + i];                                   // -tableau[0, nVars + i], :122
        sb += "  Constraint " + std::to_string(i + 1) + ": " + FormatF(shadow, 3) + "\n";
    }
    return sb;
}

static std::string trim_end(const std::string& s)
{
    size_t e = s.size();
    while (e > 0 && std::isspace((unsigned char)s[e - 1])) --e;
    return s.substr(0, e);
}
static std::string trim(const std::string& s)
{
    std::string t = trim_end(s);
    size_t b = 0; while (b < t.size() && std::isspace((unsigned char)t[b])) ++b;
    return t.substr(b);
}

SimplexResult SensitivityAnalysis::SolveUsingDuality() const                                 // :130-214
{
    const int m = (int)problem->Constraints.size(), n = problem->NumVars();
    LPProblem dual;                                   // ObjectiveSense stays at its default, Max (:156-160)
    dual.C.resize(m);
    for (int i = 0; i < m; ++i) dual.C[i] = problem->Constraints[i].B;                       // :136-138
    for (int j = 0; j < n; ++j) {                                                            // :140-152
        Constraint c;
        c.A.resize(m);
        for (int i = 0; i < m; ++i) {
            if ((int)problem->Constraints[i].A.size() <= j) throw LpxException(LPX_EINVAL, "Index was outside the bounds of the array.");
            c.A[i] = problem->Constraints[i].A[j];
        }
        c.B = problem->C[j];
        c.Relation = Rel::GE;
        dual.Constraints.push_back(std::move(c));
    }
    LPSolver solver(opt);
    std::string trace;
    UpdatePivot capture = [&trace](const std::string& text, const Highlight*) {             // :166-170
        if (!text.empty()) trace += trim_end(text) + "\n";
    };
    SimplexResult dr;
    try {
        dr = solver.Solve(dual, "Dual Simplex", capture);                                    // :176
        std::vector<std::string> t;
        if (dr.HasSolution) for (double x : dr.Solution) t.push_back(FormatF(x, 3));
        trace += "Debug: Raw Solution = [" + join(t, ", ") + "]\n";                          // :178
        trace += "Debug: VarNames = [" + (dr.HasSolution ? join(dr.VarNames, ", ") : std::string()) + "]\n";   // :179
    } catch (const LpxException& ex) {                                                       // :181-189
        if (ex.code == LPX_EDEVICE || ex.code == LPX_ENOMEM) throw;
        trace += std::string("Error solving dual LP: ") + ex.what() + "\n";
        SimplexResult err;
        err.Report = trace; err.Summary = std::string("Error: ") + ex.what();
        err.Status = LPX_CUT_ERROR;
        return err;
    }
    std::string report = "=== Duality Algorithm Solution ===\n";                            // :192-207
    {
        std::vector<std::string> t;
        for (int i = 0; i < m; ++i) t.push_back(FormatF(dual.C[i], 3) + "y" + std::to_string(i + 1));
        report += "Dual Problem: Minimize " + join(t, " + ") + "\n";
    }
    report += "Subject to:\n";
    for (const Constraint& c : dual.Constraints) {
        std::vector<std::string> t;
        for (size_t j = 0; j < c.A.size(); ++j) t.push_back(FormatF(c.A[j], 3) + "y" + std::to_string(j + 1));
        report += join(t, " + ") + " " + rel_name(c.Relation) + " " + FormatF(c.B, 3) + "\n";
    }
    report += "\n=== Duality Algorithm Iterations ===\n";
    report += trim(trace) + "\n";
    report += "\n=== Final Result ===\n";
    report += std::string("Status: ") + (dr.Summary.find("OPTIMAL") != std::string::npos ? "OPTIMAL" : "INFEASIBLE") + "\n";
    report += "w* = " + FormatF(dr.OptimalValue, 3) + "\n";
    {
        std::vector<std::string> t;
        if (dr.HasSolution) for (size_t j = 0; j < dr.Solution.size() && (int)j < n; ++j) t.push_back(FormatF(dr.Solution[j], 3));
        report += "y* = [" + join(t, ", ") + "]\n";
    }
    SimplexResult out;                                                                       // :209-218
    out.Report = report; out.Summary = dr.Summary; out.OptimalValue = dr.OptimalValue;
    out.Status = dr.Status; out.Stats = dr.Stats; out.Trace = dr.Trace;
    if (dr.HasSolution) {
        out.Solution.assign(dr.Solution.begin(), dr.Solution.begin() + std::min<size_t>(dr.Solution.size(), (size_t)n));
        out.HasSolution = true;
        out.Tableau = std::move(dr.Tableau); out.R = dr.R; out.C = dr.C;
        out.Basis = std::move(dr.Basis); out.VarNames = std::move(dr.VarNames);
    } else {
        out.Solution.assign((size_t)n, 0.0);        // `?? new double[n]`; Tableau/Basis/VarNames stay null
    }
    return out;
}

}}  // namespace lpx::host
