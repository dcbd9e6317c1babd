"""Probe: config 4 cold level search (resident group kernel): nodes/s and host phase timing (LPX_BNB_TIMING=1)."""
import sys, time, os
os.environ.setdefault("LPX_BNB_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(512, 256)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for _ in range(2):
    t0 = time.perf_counter()
    r = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=conc, max_nodes=800).Solve(pb)
    dt = time.perf_counter() - t0
    print(f"cold conc={conc}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} pivots={r.Stats['pivots']} launches={r.Stats['launches']}", flush=True)
