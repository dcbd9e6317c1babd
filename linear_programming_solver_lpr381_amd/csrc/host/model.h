// host/model.h -- C++ mirror of the reference's plugin boundary: data model, ILPAlgorithm,
// LPSolver (Models/PrimalSimplex.cs:8-49, Models/IPLAlgorithm.cs:5-8, Models/LPSolver.cs:16-76).
// Same names, argument meaning and error behaviour as the C# so that tests read like tests of the
// reference; the loops themselves run on the GPU through the C ABI (include/lpx.h).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/lpx.h"

namespace lpx { namespace host {

enum class Sense { Max, Min };          // Models/PrimalSimplex.cs:8
enum class Rel { LE, GE, EQ };          // Models/PrimalSimplex.cs:9

struct Constraint {                     // Models/PrimalSimplex.cs:11-18
    std::vector<double> A;
    Rel Relation = Rel::LE;
    double B = 0.0;
};

struct LPProblem {                      // Models/PrimalSimplex.cs:20-36
    Sense ObjectiveSense = Sense::Max;
    std::vector<double> C;
    std::vector<Constraint> Constraints;
    int NumVars() const { return (int)C.size(); }
    LPProblem Clone() const { return *this; }
};

// bool[,] highlight of the reference callback: row-major R x C flags (empty = null)
struct Highlight { int R = 0, C = 0; std::vector<uint8_t> cells; };
// Action<string, bool[,]> updatePivot (Models/IPLAlgorithm.cs:7)
using UpdatePivot = std::function<void(const std::string& text, const Highlight* highlight)>;

struct SimplexResult {                  // Models/PrimalSimplex.cs:38-49
    std::string Report, Summary;
    double OptimalValue = 0.0;
    bool HasSolution = false;           // false == Solution/Tableau/Basis/VarNames are null (defect D2)
    std::vector<double> Solution;
    std::vector<double> Tableau; int R = 0, C = 0;      // row-major R x C
    std::vector<int32_t> Basis;
    std::vector<std::string> VarNames;
    // additions of this engine (not in the reference record)
    int Status = LPX_OPTIMAL;           // LPX_OPTIMAL / LPX_UNBOUNDED / LPX_INFEASIBLE
    std::vector<int32_t> Trace;         // (leaving row, entering column) per pivot
    lpx_stats Stats{};
    int64_t LpSolves = 0, Nodes = 0;    // branch-and-bound counters
    std::vector<int32_t> NodeLog;       // B&B: (depth, outcome, branching var) per visited node
    std::vector<double> NodeZ;
    std::vector<double> Cuts;           // cutting plane: (A[0..n), B) per cut, in the order added
    std::vector<double> Aux;            // sharded B&B: {levels, all-reduces, rebalancing rounds, node descriptors moved}
};

// The reference throws System.Exception with fixed messages; `code` is the LPX_E_* of include/lpx.h.
struct LpxException : std::runtime_error {
    int code;
    LpxException(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

struct ILPAlgorithm {                   // Models/IPLAlgorithm.cs:5-8
    virtual ~ILPAlgorithm() = default;
    virtual SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) = 0;
};

// Engine knobs that have no counterpart in the reference (all default to reference behaviour).
struct EngineOptions {
    bool render_iterations = false;  // true: format the whole tableau for every pivot callback as the
                                     // reference does (Models/PrimalSimplex.cs:113-121); needs one
                                     // download per pivot.  false: callbacks get one line per pivot.
    int batch = 0;                   // pivots per host poll (0 = library default)
    int dual_flags = 0;              // 0 faithful (D1/D2 kept); LPX_DUAL_REPAIRED = repaired
    int bnb_mode = 0;                // 0 faithful, 1 repaired
    int bnb_search = 0;              // 0 = reference DFS (ceil first), 1 = level-synchronous sharded
    int64_t max_nodes = 0;           // 0 = unlimited; sharded searches: budget of the WHOLE job (split over the ranks)
    int concurrent_nodes = 1;        // level-synchronous search: node LPs in flight per GPU
    int rank = 0, world = 1;         // level-synchronous search: shard of this process
    // incumbent exchange: called once per level with {best_z, have_work}; must return the MAX over
    // ranks in place (RCCL all-reduce in production, identity for one process).
    std::function<void(double* vals, int count)> allreduce_max;
    bool shard_one = false;          // diagnostic (LPX_COMM_SHARD_ONE): run the sharded code path with world == 1
    int max_iter = 10000;
    int bnb_dive = 0;                // sharded searches: 1 = depth-first-K pool policy
    bool quiet = false;              // internal solves whose Report nobody reads (B&B nodes): skip the canonical-form text,
                                     // hundreds of thousands of formatted numbers for a config-4 model
    // test seams (see include/lpx_test.h); never set by product code
    int64_t test_fail_after_nodes = 0;
    std::function<int(double* T, int R, int C, int32_t* basis, int dual, int repaired, int max_iter, int nvars,
                      double* x, double* z, int64_t* pivots)> test_node_lp;
    std::function<int(int count, const int32_t* off, const int32_t* fidx, const int8_t* fval, double* profit,
                      double* weight, int32_t* frac, double* fracval)> test_knap_relax;
};

enum { LPX_DUAL_FIX_D1 = 1, LPX_DUAL_FIX_D2 = 2, LPX_DUAL_SOUND = 4, LPX_DUAL_REPAIRED = 7 };

class PrimalSimplex : public ILPAlgorithm {             // Models/PrimalSimplex.cs:52-305
public:
    explicit PrimalSimplex(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& original, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class DualSimplex : public ILPAlgorithm {               // Models/DualSimplex.cs:11-313
public:
    explicit DualSimplex(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& original, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class RevisedPrimalSimplex : public ILPAlgorithm {      // Models/RevisedPrimalSimplex.cs:12-458
public:
    explicit RevisedPrimalSimplex(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& original, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class BranchAndBound : public ILPAlgorithm {            // Models/Branch&Bound.cs:20-304
public:
    explicit BranchAndBound(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) override;
    double BestObjective = -1.0 / 0.0;
    std::vector<double> BestSolution; bool HasBest = false;
private:
    EngineOptions opt;
};

class BranchAndBoundRevised : public ILPAlgorithm {     // Models/BranchAndBoundRevised.cs:17-390
public:
    explicit BranchAndBoundRevised(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class BranchAndBoundKnapsack : public ILPAlgorithm {    // Models/BranchAndBoundKnapsack.cs:12-548
public:
    explicit BranchAndBoundKnapsack(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class CuttingPlane : public ILPAlgorithm {              // Models/CuttingPlane.cs:9-164
public:
    explicit CuttingPlane(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

class CuttingPlaneRevised : public ILPAlgorithm {       // Models/CuttingPlaneRevised.cs:9-112
public:
    explicit CuttingPlaneRevised(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, UpdatePivot updatePivot = nullptr) override;
private:
    EngineOptions opt;
};

// Models/SensitivityAnalysis.cs:11-297.  `problem` is mutated by ApplyChange, as in the
// reference; `result` must carry Tableau/Basis/VarNames (the constructor checks, :24-43).
class SensitivityAnalysis {
public:
    SensitivityAnalysis(LPProblem* problem, const SimplexResult* result, const EngineOptions& o = {});
    std::string GetRangeReport(const std::string& target) const;          // :47-76
    std::string ApplyChange(const std::string& target, double value);     // :78-107
    std::string GetShadowPricesReport() const;                            // :109-128
    SimplexResult SolveUsingDuality() const;                              // :130-219
    std::pair<double, double> Range(const std::string& target) const;     // GetRangeReport's numbers
    void Locate(const std::string& target, int* field, int* index) const;  // entry ApplyChange assigns
    std::pair<double, double> GetConstraintRange(int index) const;                    // :277-298
    std::pair<double, double> GetBasicVariableObjectiveRange(int basicVarRow) const;  // :250-275
    std::pair<double, double> GetNonBasicVariableRange(const std::string& varName) const;   // :229-248
private:
    bool basis_contains(int col) const;
    int constraint_index(const std::string& target) const;
    int var_column(const std::string& target) const;
    LPProblem* problem; const SimplexResult* result; EngineOptions opt;
};

class LPSolver {                                        // Models/LPSolver.cs:6-77
public:
    explicit LPSolver(const EngineOptions& o = {}) : opt(o) {}
    SimplexResult Solve(const LPProblem& problem, const std::string& algorithm, UpdatePivot updatePivot = nullptr);
    static std::string NormalizeAlgorithmKey(const std::string& algorithm);   // :61-76
    std::vector<double> FinalTableau; int FinalR = 0, FinalC = 0; bool HasFinalTableau = false;   // :11
private:
    EngineOptions opt;
};

// LPParser.ParseFromText, Models/LPParser.cs:9-79.  Throws LpxException(LPX_E_PARSE, message).
LPProblem ParseFromText(const std::string& input);

// Text formats of the reference (Models/PrimalSimplex.cs:259-304 and friends), see format.cpp
std::string FormatNumber(double v);                 // ToString("0.###")
std::string FormatF(double v, int decimals);        // ToString("F3"/"F6", InvariantCulture)
std::string FormatRound3(double v);                 // $"{Math.Round(v, 3):0.###}"
std::string FormatShortest(double v);               // double.ToString(): shortest round-trippable digits
double RoundHalfEven(double v, int decimals);       // Math.Round(v, decimals)
std::string AppendTableau(const std::string& title, const double* T, int R, int C,
                          const std::vector<int32_t>& basis, const std::vector<std::string>& varNames, int iter);
std::string AppendCanonicalForm(const LPProblem& model);
std::string MatrixToString(const double* M, int r, int c);
std::string BuildIterationBlock(int iter, const std::vector<int32_t>& Bidx, const std::vector<int32_t>& Nidx,
                                const std::vector<std::string>& names, const double* Binv, int m,
                                const std::vector<double>& xB, double z, const std::vector<double>* rN,
                                int entering, const std::vector<double>* d, double bestTheta, double eps);

// helpers shared by the solver mirrors (solvers.cpp)
void BuildTableauPrimal(const LPProblem& expanded, std::vector<double>& T, int& R, int& C,
                        std::vector<int32_t>& basis, std::vector<std::string>& varNames);
LPProblem ExpandEqualitiesToInequalities(const LPProblem& model);
LPProblem PrepareForTableauDual(const LPProblem& original, bool fix_d1);

// Handles of whole-model solves and of the B&B root templates, kept from one solve to the next (creating a handle costs ~4 ms of device
// and pinned allocations, destroying it ~3 ms: two root solves and two templates were 25-30 ms of every search).  Exact shapes only --
// these handles are never rebuilt to another size -- small tableaux only (<= 32 MB), a bounded number; everything else is created and
// destroyed as before.  acquire: a cached handle of that shape or a new one (throws LpxException); release: back to the cache or destroyed.
::lpx_tableau* acquire_exact_handle(int R, int C);
void release_exact_handle(::lpx_tableau* t, int R, int C);

}
}  // namespace lpx::host
