"""Probe: config 4 sharded level search with the depth-first-K pool: does an incumbent appear, at what node count?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(512, 256)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
for search, conc, budget, dive in [(1, 64, 400, 0), (1, 16, 1200, 1), (1, 64, 2400, 1), (2, 64, 4000, 0), (2, 16, 4000, 1), (2, 64, 8000, 1), (2, 8, 4000, 1)]:
    t0 = time.perf_counter()
    r = L.BranchAndBound(bnb_mode=1, bnb_search=search, bnb_dive=dive, concurrent_nodes=conc, max_nodes=budget).Solve(pb)
    dt = time.perf_counter() - t0
    depth = int(r.NodeLog[:, 0].max()) if len(r.NodeLog) else -1
    outc = np.bincount(r.NodeLog[:, 1], minlength=9).tolist() if len(r.NodeLog) else []
    print(f"search={search} conc={conc} budget={budget} dive={dive}: {dt:.2f}s lp={r.LpSolves} nodes/s={r.LpSolves/dt:.0f} incumbent={r.OptimalValue} maxdepth={depth} outcomes={outc} aux={r.Aux}", flush=True)
