"""Probe: config 4 warm-started sharded search (batched streaming dual kernels): nodes/s."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(512, 256)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
conc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
budget = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    t0 = time.perf_counter()
    r = L.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=conc, max_nodes=budget).Solve(pb)
    dt = time.perf_counter() - t0
    print(f"warm conc={conc} budget={budget}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} pivots={r.Stats['pivots']}", flush=True)
