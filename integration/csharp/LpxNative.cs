// LpxNative.cs -- P/Invoke declarations of liblpx.so (include/lpx.h), for the reference application
// Jellyman750/Linear_Programming_Solver_LPR381.  Shipped as source: no .NET toolchain exists in the build image, so this
// file has NOT been compiled here; every entry point below is exercised through the same C ABI by tests/test_gpu_*.py.
//
// Layouts mirror include/lpx.h field by field (LayoutKind.Sequential, cdecl).  Nothing from include/lpx_test.h appears here.
using System;
using System.Runtime.InteropServices;

namespace Linear_Programming_Solver.Native
{
    [StructLayout(LayoutKind.Sequential)]
    public struct LpxStats                       // lpx_stats
    {
        public long pivots, launches;
        public double loop_ms, h2d_ms, d2h_ms, update_ms_sum;
        public long update_launches, fdf_pivots, cleanup_pivots;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct LpxProblem              // lpx_problem  (LPProblem, Models/PrimalSimplex.cs:20-36)
    {
        public int sense, n, m;                  // sense: 0 = Max, 1 = Min (enum Sense, :8)
        public double* c;                        // [n]
        public double* A;                        // [m*n] row-major
        public int* rel;                         // [m]   0 = LE, 1 = GE, 2 = EQ (enum Rel, :9)
        public double* b;                        // [m]
    }

    [UnmanagedFunctionPointer(CallingConvention.Cdecl)]
    public delegate void LpxPivotCb(IntPtr user, int iter, int row, int col);
    [UnmanagedFunctionPointer(CallingConvention.Cdecl)]
    public unsafe delegate void LpxAllreduceMax(IntPtr user, double* vals, int count);
    [UnmanagedFunctionPointer(CallingConvention.Cdecl)]
    public unsafe delegate void LpxTextCb(IntPtr user, byte* text, byte* highlight, int R, int C);

    [StructLayout(LayoutKind.Sequential)]
    public struct LpxSolveOpts                   // lpx_solve_opts
    {
        public int max_iter, batch, render_iterations, dual_flags, bnb_mode, bnb_search, concurrent_nodes, rank, world;
        public long max_nodes;
        public IntPtr allreduce_max, allreduce_user;     // LpxAllreduceMax via Marshal.GetFunctionPointerForDelegate
        public IntPtr text_cb, text_user;                // LpxTextCb
        public int bnb_dive;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct LpxResult               // lpx_result  (SimplexResult, Models/PrimalSimplex.cs:38-49)
    {
        public int status, has_solution;
        public double optimal_value;
        public int n; public double* x;
        public int R, C; public double* T;
        public int* basis;
        public int n_pivots; public int* trace;
        public byte* report; public byte* summary;
        public long lp_solves, nodes;
        public int n_log; public int* node_log; public double* node_z;
        public fixed double aux[4];
        public LpxStats stats;
        public int n_cuts; public double* cuts;
    }

    public static unsafe class Lpx
    {
        const string Lib = "lpx";                // liblpx.so (Linux) next to the executable / on LD_LIBRARY_PATH

        public const int OPTIMAL = 0, UNBOUNDED = 1, INFEASIBLE = 2, ITER_LIMIT = 3;
        public const int E_GE_PRESENT = -10, E_NEG_RHS = -11, E_REVISED_PRECOND = -12, E_SINGULAR = -13,
                         E_KNAP_SHAPE = -14, E_UNKNOWN_ALGO = -15, E_PARSE = -16;

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_abi_version();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_device_count();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_init(int device);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_last_error(byte[] buf, int len);

        // ---- X1: the incumbent exchange of a sharded search, one process per GPU (include/lpx.h lpx_comm_*): the library owns an
        //      RCCL communicator; BestObjective (Models/Branch&Bound.cs:182,191) / _bestValue (Models/BranchAndBoundKnapsack.cs:124)
        //      become one ncclAllReduce(ncclMax, ncclDouble) per level / round.  lpx_init(localGpu) first.
        public const int COMM_ID_BYTES = 128;
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_comm_unique_id(byte[] id);          // rank 0
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_comm_init(int rank, int world, byte[] id);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_comm_init_tcp(int rank, int world, [MarshalAs(UnmanagedType.LPUTF8Str)] string host, int port);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_comm_allreduce_max(double* vals, int count);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_comm_info(out int rank, out int world, out long allreduces, out double allreduceMs, out int rcclVersion);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int lpx_comm_destroy();

        // ---- loop-level entry points: the reference keeps its own model preparation and reports (INTEGRATION.md section 2) ----
        // replaces the while(true) of PrimalSimplex.Solve, Models/PrimalSimplex.cs:92-124
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_primal_tableau(double* T, int R, int C, int* basis, double eps, int maxIter,
                                                    LpxPivotCb cb, IntPtr user, out LpxStats st);
        // replaces ForceDualFeasibility + the loop of DualSimplex.Solve, Models/DualSimplex.cs:24,36-113
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_dual_tableau(double* T, int R, int C, int* basis, double eps, double ratioTol,
                                                  int fdfGuard, int maxIter, int cleanup, LpxPivotCb cb, IntPtr user, out LpxStats st);
        // replaces the for-loop of RevisedPrimalSimplex.Solve, Models/RevisedPrimalSimplex.cs:58-142
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_revised_solve(double* A, int m, int n, double* c, double* b, int* Bidx, int* Nidx,
                                                   double* xB, out double z, double eps, int maxIter,
                                                   LpxPivotCb cb, IntPtr user, out LpxStats st);
        // batched ComputeRelaxation, Models/BranchAndBoundKnapsack.cs:431-491
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_knapsack_create(double* profit, double* weight, int n, double cap, out IntPtr k);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern void lpx_knapsack_destroy(IntPtr k);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_knapsack_relax_batch(IntPtr k, int count, int* off, int* fixIdx, sbyte* fixVal,
                                                          double* profit, double* weight, int* fracIdx, double* fracVal);

        // ---- model-level entry point: ILPAlgorithm.Solve as dispatched by LPSolver.Solve (Models/LPSolver.cs:16-59) ----
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern void lpx_default_solve_opts(out LpxSolveOpts o);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)]
        public static extern int lpx_solve(ref LpxProblem p, [MarshalAs(UnmanagedType.LPUTF8Str)] string algorithm,
                                           ref LpxSolveOpts o, out LpxResult result);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern void lpx_result_free(ref LpxResult r);

        public static string LastError()
        {
            var b = new byte[1024];
            lpx_last_error(b, b.Length);
            int len = Array.IndexOf(b, (byte)0);
            return System.Text.Encoding.UTF8.GetString(b, 0, len < 0 ? b.Length : len);
        }
    }
}
