"""Probe: small prunable 0/1 IPs solved to optimality by the sharded level search (incumbent found, bound prunes)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
for (n, m) in [(30, 8), (40, 10), (50, 10), (60, 12)]:
    cb, Ab, relb, bb = synth.binary_ip(n, m)
    pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
    for search, conc, dive in [(1, 64, 0), (1, 64, 1), (2, 64, 1)]:
        t0 = time.perf_counter()
        r = L.BranchAndBound(bnb_mode=1, bnb_search=search, bnb_dive=dive, concurrent_nodes=conc, max_nodes=40000).Solve(pb)
        dt = time.perf_counter() - t0
        outc = np.bincount(r.NodeLog[:, 1], minlength=9).tolist() if len(r.NodeLog) else []
        print(f"n={n} m={m} search={search} dive={dive}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} z={r.OptimalValue} outcomes={outc} aux={r.Aux}", flush=True)
