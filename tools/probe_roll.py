import sys, time, os
sys.path.insert(0, "/root/repo")
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(512, 256)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
for k in range(2):
    t0 = time.perf_counter(); r = L.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=int(os.environ.get("CONC","64")), max_nodes=8000).Solve(pb); dt = time.perf_counter() - t0
print(f"div={os.environ.get('LPX_ROLL_DIV')} batch={os.environ.get('LPX_ROLL_BATCH')} conc={os.environ.get('CONC')}: {dt:.2f}s nodes/s={r.LpSolves/dt:.0f} pivots={r.Stats['pivots']}", flush=True)
