"""Probe: config 5 (100k-item knapsack) best-first search, nodes/s and launches."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
pk, wk, capk = synth.knapsack(100_000)
kp = L.LPProblem(L.Sense.Max, pk.tolist(), [L.Constraint(wk.tolist(), L.Rel.LE, capk)])
for conc, budget in ((512, 1), (64, 200000), (512, 200000), (512, 1000000), (2048, 1000000), (8192, 1000000)):
    t0 = time.perf_counter(); r = L.BranchAndBoundKnapsack(max_nodes=budget, concurrent_nodes=conc).Solve(kp); dt = time.perf_counter() - t0
    print(f"conc={conc}: {r.Nodes} pops {dt:.3f} s, {r.Nodes/dt:.0f} nodes/s, launches {r.Stats['launches']}, {1e6*dt/max(r.Stats['launches'],1):.1f} us per launch, device calls {r.Stats['loop_ms']:.1f} ms of {1e3*dt:.1f}, z {r.OptimalValue}", flush=True)
