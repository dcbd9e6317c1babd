// LpxAlgorithms.cs -- drop-in ILPAlgorithm implementations backed by liblpx.so (MI355X).  Register them in
// LPSolver.Solve's switch (Models/LPSolver.cs:20-43) in place of `new PrimalSimplex()` etc., or construct them directly as
// Form1.btnSolve_Click does (Form1.cs:244-268).  Shipped as source, not compiled here (no .NET toolchain in the build image).
//
// Each class hands the LPProblem to lpx_solve under the reference's own algorithm name and rebuilds a SimplexResult from
// what comes back: Report / Summary text in the reference's format, OptimalValue, Solution, Tableau (double[R,C], objective
// row last), Basis and VarNames -- or nulls where the reference returns nulls (DualSimplex, RevisedPrimalSimplex: text only).
// Exceptions carry the reference's messages (LPX_E_* codes, include/lpx.h).
using System;
using System.Linq;
using System.Runtime.InteropServices;
using Linear_Programming_Solver.Native;

namespace Linear_Programming_Solver.Models
{
    public class LpxAlgorithm : ILPAlgorithm
    {
        private readonly string _name;
        public int DualFlags;        // 0 = the reference's DualSimplex (defects D1/D2 kept), 7 = repaired
        public int BnbMode;          // 0 = faithful, 1 = repaired
        public int BnbSearch;        // 0 = the reference's DFS, 1 = level-synchronous (shardable), 2 = warm-started children
        public int ConcurrentNodes = 1;
        public bool RenderIterations;    // true: the callback receives the whole tableau text per pivot, as the reference does

        public LpxAlgorithm(string referenceAlgorithmName) { _name = referenceAlgorithmName; }

        public unsafe SimplexResult Solve(LPProblem problem, Action<string, bool[,]> updatePivot = null)
        {
            if (problem == null) throw new ArgumentNullException(nameof(problem));
            int n = problem.NumVars, m = problem.Constraints.Count;
            double[] c = (double[])problem.C.Clone();
            var A = new double[Math.Max(1, m * n)];
            var rel = new int[Math.Max(1, m)];
            var b = new double[Math.Max(1, m)];
            for (int i = 0; i < m; i++)
            {
                var row = problem.Constraints[i];
                if (row.A.Length < n) throw new IndexOutOfRangeException();        // what BuildTableau would hit, Models/PrimalSimplex.cs:190
                Array.Copy(row.A, 0, A, i * n, n);
                rel[i] = (int)row.Relation;
                b[i] = row.B;
            }
            Lpx.lpx_default_solve_opts(out var o);
            o.dual_flags = DualFlags; o.bnb_mode = BnbMode; o.bnb_search = BnbSearch;
            o.concurrent_nodes = ConcurrentNodes; o.render_iterations = RenderIterations ? 1 : 0;
            LpxTextCb textCb = null;
            if (updatePivot != null)
            {
                textCb = (user, text, hl, R, C) =>
                {
                    string s = Marshal.PtrToStringUTF8((IntPtr)text);
                    bool[,] mask = null;
                    if (hl != null && R > 0 && C > 0)
                    {
                        mask = new bool[R, C];
                        for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) mask[i, j] = hl[i * C + j] != 0;
                    }
                    updatePivot(s, mask);           // same thread, between iterations (Form1.AppendPivotRow touches a RichTextBox)
                };
                o.text_cb = Marshal.GetFunctionPointerForDelegate(textCb);
            }
            LpxResult r;
            int rc;
            fixed (double* pc = c, pA = A, pb = b)
            fixed (int* prel = rel)
            {
                var p = new LpxProblem { sense = (int)problem.ObjectiveSense, n = n, m = m, c = pc, A = pA, rel = prel, b = pb };
                rc = Lpx.lpx_solve(ref p, _name, ref o, out r);
            }
            GC.KeepAlive(textCb);
            if (rc != 0) throw new Exception(Lpx.LastError());       // the reference's own message for its own exceptions
            try
            {
                var res = new SimplexResult
                {
                    Report = Marshal.PtrToStringUTF8((IntPtr)r.report) ?? "",
                    Summary = Marshal.PtrToStringUTF8((IntPtr)r.summary) ?? "",
                    OptimalValue = r.optimal_value
                };
                if (r.has_solution != 0)
                {
                    res.Solution = new double[r.n];
                    for (int j = 0; j < r.n; j++) res.Solution[j] = r.x[j];
                    if (r.R > 0 && r.C > 0)
                    {
                        res.Tableau = new double[r.R, r.C];
                        fixed (double* dst = res.Tableau) Buffer.MemoryCopy(r.T, dst, 8L * r.R * r.C, 8L * r.R * r.C);
                        res.Basis = new int[r.R - 1];
                        for (int i = 0; i < r.R - 1; i++) res.Basis[i] = r.basis[i];
                        int ns = r.C - 1 - n;                          // x1..xn, c1..cm as BuildTableau names them (:200-201)
                        res.VarNames = Enumerable.Range(1, n).Select(j => "x" + j)
                                                 .Concat(Enumerable.Range(1, Math.Max(0, ns)).Select(j => "c" + j)).ToArray();
                    }
                }
                return res;
            }
            finally { Lpx.lpx_result_free(ref r); }
        }
    }

    // The names Form1's dropdown and LPSolver's switch use (Form1.cs:64-69, Models/LPSolver.cs:20-36)
    public sealed class PrimalSimplexLpx : LpxAlgorithm { public PrimalSimplexLpx() : base("Primal Simplex") { } }
    public sealed class RevisedPrimalSimplexLpx : LpxAlgorithm { public RevisedPrimalSimplexLpx() : base("Revised Primal Simplex") { } }
    public sealed class DualSimplexLpx : LpxAlgorithm { public DualSimplexLpx() : base("Dual Simplex") { } }
    public sealed class BranchAndBoundLpx : LpxAlgorithm { public BranchAndBoundLpx() : base("Branch and Bound") { } }
    public sealed class BranchAndBoundRevisedLpx : LpxAlgorithm { public BranchAndBoundRevisedLpx() : base("Revised Branch and Bound") { } }
    public sealed class BranchAndBoundKnapsackLpx : LpxAlgorithm { public BranchAndBoundKnapsackLpx() : base("Branch and Bound Knapsack") { } }
    public sealed class CuttingPlaneLpx : LpxAlgorithm { public CuttingPlaneLpx() : base("Cutting Plane") { } }
    public sealed class CuttingPlaneRevisedLpx : LpxAlgorithm { public CuttingPlaneRevisedLpx() : base("Revised Cutting Plane") { } }
}
