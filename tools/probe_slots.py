import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(60, 12)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
for conc in (32, 64, 128, 240):
    for _ in range(2):
        t0 = time.perf_counter()
        r = L.BranchAndBound(bnb_mode=1, bnb_search=1, bnb_dive=1, concurrent_nodes=conc, max_nodes=0).Solve(pb)
        dt = time.perf_counter() - t0
    print(f"slots env={os.environ.get('LPX_GROUP_SLOTS')} conc={conc}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} z={r.OptimalValue} pivots={r.Stats['pivots']}", flush=True)

for conc in (64, 128):
    for _ in range(2):
        t0 = time.perf_counter()
        r = L.BranchAndBound(bnb_mode=1, bnb_search=2, bnb_dive=1, concurrent_nodes=conc, max_nodes=0).Solve(pb)
        dt = time.perf_counter() - t0
    print(f"WARM conc={conc}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} z={r.OptimalValue} pivots={r.Stats['pivots']}", flush=True)
