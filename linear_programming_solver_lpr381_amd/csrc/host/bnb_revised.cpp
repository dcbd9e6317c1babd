// host/bnb_revised.cpp -- BranchAndBoundRevised mirror (Models/BranchAndBoundRevised.cs:17-390).
//
// The reference solves every node through LPSolver ("Revised Primal Simplex" when all rows are `<=`, else
// "Dual Simplex", :238-243) and then READS x* AND z* BACK FROM THE SUMMARY TEXT (:276-389).  The mirror does
// literally that: it calls the RevisedPrimalSimplex / DualSimplex mirrors (which run on the GPU and render
// the reference's Summary format) and re-parses their text with restatements of ParseSolutionVector,
// ParseSolutionVectorFromTableau and ParseObjectiveValue -- so the three-decimal quantisation of the
// reference is reproduced by construction.
#include "model.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace lpx { namespace host {

namespace {

constexpr double EPS = 1e-6;      // :21
constexpr int MaxDepth = 200;     // :22
enum Outcome { O_ERROR = 0, O_INVALID = 1, O_INFEASIBLE_X = 2, O_PRUNED = 3, O_INCUMBENT = 4, O_NO_FRAC = 5,
               O_BRANCHED = 6, O_DEPTH = 7 };

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// double.Parse(s, NumberStyles.Any, InvariantCulture) for the strings the summaries contain
bool parse_num(std::string s, double& v)
{
    s = trim(s);
    for (char& ch : s) if (ch == ',') ch = '.';
    if (s.empty()) return false;
    if (s == "\xE2\x88\x9E") { v = INFINITY; return true; }
    if (s == "-\xE2\x88\x9E") { v = -INFINITY; return true; }
    if (s == "NaN") { v = NAN; return true; }
    char* e = nullptr;
    v = std::strtod(s.c_str(), &e);
    return e != s.c_str() && *e == 0;
}

// ParseSolutionVector, :276-323
std::vector<double> ParseSolutionVector(const std::string& summary, int expected)
{
    size_t start = summary.find("x* = [");
    if (start == std::string::npos) return {};
    start += 6;
    size_t end = summary.find("]", start);
    if (end == std::string::npos) return {};
    std::string vec = trim(summary.substr(start, end - start));
    if (vec.empty()) return {};
    std::vector<std::string> parts;
    for (size_t p = 0;;) {                                   // Split(", ", RemoveEmptyEntries)
        size_t q = vec.find(", ", p);
        std::string part = vec.substr(p, q == std::string::npos ? std::string::npos : q - p);
        if (!part.empty()) parts.push_back(part);
        if (q == std::string::npos) break;
        p = q + 2;
    }
    std::vector<double> values(expected, 0.0);
    for (int i = 0; i < std::min((int)parts.size(), expected); ++i)
        if (!parse_num(parts[i], values[i])) return {};      // FormatException -> catch -> empty
    if ((int)parts.size() != expected) return {};
    return values;
}

// ParseSolutionVectorFromTableau, :325-369 (lines "x3 ... <rhs>"; a Summary has none, so this yields zeros)
std::vector<double> ParseSolutionVectorFromTableau(const std::string& summary, int numVars)
{
    std::vector<double> values(numVars, 0.0);
    size_t p = 0;
    while (p <= summary.size()) {
        size_t e = summary.find_first_of("\r\n", p);
        std::string line = trim(summary.substr(p, e == std::string::npos ? std::string::npos : e - p));
        if (!line.empty() && (line[0] == 'x' || line[0] == 'v')) {
            std::vector<std::string> parts;
            for (size_t a = 0; a < line.size();) {
                while (a < line.size() && std::isspace((unsigned char)line[a])) ++a;
                size_t b = a; while (b < line.size() && !std::isspace((unsigned char)line[b])) ++b;
                if (b > a) parts.push_back(line.substr(a, b - a));
                a = b;
            }
            if (parts.size() >= 2 && parts[0].size() >= 2) {
                const std::string idx = parts[0].substr(1);
                bool digits = !idx.empty(); for (char ch : idx) if (!std::isdigit((unsigned char)ch)) digits = false;
                if (digits) {
                    int vi = std::atoi(idx.c_str());
                    double rhs;
                    if (vi >= 1 && vi <= numVars && parse_num(parts.back(), rhs)) values[vi - 1] = rhs;
                }
            }
        }
        if (e == std::string::npos) break;
        p = e + 1;
    }
    return values;
}

// ParseObjectiveValue, :371-388
double ParseObjectiveValue(const std::string& summary)
{
    size_t p = 0;
    while (p <= summary.size()) {
        size_t e = summary.find('\n', p);
        std::string line = trim(summary.substr(p, e == std::string::npos ? std::string::npos : e - p));
        if (line.rfind("z*", 0) == 0) {
            size_t eq = line.find('=');
            if (eq != std::string::npos && line.find('=', eq + 1) == std::string::npos) {
                double v;
                if (parse_num(line.substr(eq + 1), v)) return v;
            }
        }
        if (e == std::string::npos) break;
        p = e + 1;
    }
    return -INFINITY;
}

struct Cut { int var; Rel rel; double bound; };

struct Ctx {
    const LPProblem* root; EngineOptions opt; UpdatePivot cb; SimplexResult* out;
    double best = -INFINITY; bool has_best = false; std::vector<double> best_x; bool stop = false;
    void log(const std::string& s) { if (cb) cb(s + "\n", nullptr); }
};

LPProblem make_node(const LPProblem& root, const std::vector<Cut>& cuts)
{
    LPProblem p = root.Clone();
    for (const Cut& c : cuts) {
        Constraint k; k.A.assign(root.NumVars(), 0.0); k.A[c.var] = 1.0; k.Relation = c.rel; k.B = c.bound;
        p.Constraints.push_back(k);
    }
    return p;
}

const char* ChooseAlgorithm(const LPProblem& p)                  // :238-243
{
    for (const Constraint& c : p.Constraints) if (c.Relation == Rel::GE || c.Relation == Rel::EQ) return "Dual Simplex";
    return "Revised Primal Simplex";
}

bool IsIntegral(const std::vector<double>& x)                    // :245-248
{
    for (double v : x) if (!(std::fabs(v - std::nearbyint(v)) < EPS)) return false;
    return true;
}

bool IsFeasible(const std::vector<double>& x, const LPProblem& p)   // :250-266
{
    for (const Constraint& c : p.Constraints) {
        double sum = 0;
        for (size_t i = 0; i < x.size(); ++i) sum += c.A[i] * x[i];
        if (c.Relation == Rel::LE && sum > c.B + EPS) return false;
        if (c.Relation == Rel::GE && sum < c.B - EPS) return false;
        if (c.Relation == Rel::EQ && std::fabs(sum - c.B) > EPS) return false;
    }
    for (double v : x) if (!(v >= -EPS)) return false;
    return true;
}

// solve + parse (:120-141).  0 ok, 1 exception, 3 unparsable vector
int solve_and_parse(Ctx& c, const LPProblem& p, std::vector<double>& x, double& z)
{
    EngineOptions lo = c.opt; lo.dual_flags = c.opt.bnb_mode == 1 ? LPX_DUAL_REPAIRED : 0;
    LPSolver solver(lo);
    SimplexResult res;
    c.out->LpSolves++;
    try {
        res = solver.Solve(p, ChooseAlgorithm(p), nullptr);
    } catch (const LpxException& ex) {
        if (ex.code == LPX_EDEVICE || ex.code == LPX_ENOMEM) throw;
        return 1;
    }
    c.out->Stats.pivots += res.Stats.pivots;
    const int n = p.NumVars();
    x = ParseSolutionVector(res.Summary, n);
    bool allz = true; for (double v : x) if (!(std::fabs(v) < EPS)) allz = false;
    if (x.empty() || allz) x = ParseSolutionVectorFromTableau(res.Summary, n);
    if ((int)x.size() != n) return 3;
    z = ParseObjectiveValue(res.Summary);
    return 0;
}

void node_log(Ctx& c, int depth, int outcome, int var, double z)
{
    c.out->NodeLog.push_back(depth); c.out->NodeLog.push_back(outcome); c.out->NodeLog.push_back(var);
    c.out->NodeZ.push_back(z);
}

void SolveSubproblem(Ctx& c, std::vector<Cut>& cuts, int depth)  // :100-234
{
    if (c.stop) return;
    if (c.opt.max_nodes > 0 && c.out->Nodes >= c.opt.max_nodes) { c.stop = true; return; }
    c.out->Nodes++;
    if (depth > MaxDepth) { node_log(c, depth, O_DEPTH, -1, 0.0); return; }
    LPProblem p = make_node(*c.root, cuts);
    std::vector<double> x; double z = 0;
    int rc = solve_and_parse(c, p, x, z);
    if (rc == 1 || rc == 3) { node_log(c, depth, O_ERROR, -1, 0.0); return; }
    if (std::isnan(z) || std::isinf(z)) { node_log(c, depth, O_INVALID, -1, z); return; }            // :143-147
    if (!IsFeasible(x, p)) { node_log(c, depth, O_INFEASIBLE_X, -1, z); return; }                      // :152-156
    if (z <= (c.has_best ? c.best : -INFINITY) + EPS) { node_log(c, depth, O_PRUNED, -1, z); return; } // :159-163
    if (IsIntegral(x)) {                                                                               // :166-172
        c.best = z; c.has_best = true; c.best_x.resize(x.size());
        for (size_t i = 0; i < x.size(); ++i) c.best_x[i] = std::nearbyint(x[i]);
        node_log(c, depth, O_INCUMBENT, -1, z);
        return;
    }
    int k = -1; double minDist = 1.7976931348623157e308;                                               // :175-190
    for (int i = 0; i < (int)x.size(); ++i) {
        double fp = x[i] - std::floor(x[i]);
        if (fp > EPS && (1 - fp) > EPS) {
            double dist = std::fabs(fp - 0.5);
            if (dist < minDist || (dist == minDist && i < k)) { minDist = dist; k = i; }
        }
    }
    if (k == -1) { node_log(c, depth, O_NO_FRAC, -1, z); return; }
    const double fl = std::floor(x[k]), ce = std::ceil(x[k]);                                          // :198-200
    node_log(c, depth, O_BRANCHED, k, z);
    cuts.push_back({k, Rel::GE, ce});
    SolveSubproblem(c, cuts, depth + 1);                                                               // ceil first, :232
    cuts.back() = {k, Rel::LE, fl};
    SolveSubproblem(c, cuts, depth + 1);                                                               // :233
    cuts.pop_back();
}

}  // namespace

SimplexResult BranchAndBoundRevised::Solve(const LPProblem& problem, UpdatePivot updatePivot)
{
    SimplexResult out;
    Ctx c; c.root = &problem; c.opt = opt; c.cb = updatePivot; c.out = &out;
    c.log("Revised Branch & Bound: starting at Root Problem.");
    std::vector<double> xRoot; double zRoot = 0;
    int rc = solve_and_parse(c, problem, xRoot, zRoot);                                                // :40-66
    if (rc == 1) { out.Report = "LP relaxation infeasible"; out.Summary = ""; out.Status = LPX_INFEASIBLE; return out; }
    if (rc == 3) { out.Report = "Failed to parse root solution"; out.Summary = ""; out.Status = LPX_INFEASIBLE; return out; }
    if (IsIntegral(xRoot) && IsFeasible(xRoot, problem)) {                                             // :69-74
        c.best = zRoot; c.has_best = true; c.best_x.resize(xRoot.size());
        for (size_t i = 0; i < xRoot.size(); ++i) c.best_x[i] = std::nearbyint(xRoot[i]);
    } else {
        std::vector<Cut> cuts;
        SolveSubproblem(c, cuts, 0);                                                                   // :78
    }
    std::string sb = "Revised Branch & Bound Finished.\n";                                             // :81-97
    if (!c.has_best) sb += "No integer-feasible solution found.\n";
    else {
        sb += "Best integer z* = " + FormatNumber(c.best) + "\n";
        sb += "Best integer x* = [";
        for (size_t i = 0; i < c.best_x.size(); ++i) { if (i) sb += ", "; sb += FormatNumber(c.best_x[i]); }
        sb += "]\n";
    }
    out.Report = sb; out.Summary = sb;
    out.Status = c.has_best ? LPX_OPTIMAL : LPX_INFEASIBLE;
    // the reference returns text only (:92-96); engine extras:
    out.HasSolution = false;
    out.OptimalValue = c.has_best ? c.best : -INFINITY;
    out.Solution = c.best_x;
    return out;
}

}}  // namespace lpx::host
