"""Probe: which mid-size 0/1 IP (node tableau larger than one CU's LDS) is SOLVED by the cold depth-first-K level search in a few
seconds with incumbents and pruning -- the `bnb_prune_mid` leg of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cases = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(96, 24, 0), (128, 32, 0), (128, 48, 0), (160, 40, 0), (192, 48, 0)]
for n, m, sd in cases:
    c, A, rel, b = synth.binary_ip(n, m, seed=synth.SEED + sd)
    p = L.LPProblem.from_arrays(0, c, A, rel, b)
    for K in (64,):
        t0 = time.perf_counter()
        r = L.BranchAndBound(bnb_mode=1, bnb_search=1, bnb_dive=1, concurrent_nodes=K, max_nodes=60000).Solve(p)
        dt = time.perf_counter() - t0
        log = np.asarray(r.NodeLog).reshape(-1, 3)
        print(f"n={n} m={m} seed+{sd} K={K}: {dt:.2f} s, LPs {r.LpSolves}, nodes {r.Nodes}, z {r.OptimalValue}, pruned {(log[:,1]==3).sum()}, "
              f"incumbents {(log[:,1]==4).sum()}, pivots {r.Stats['pivots']}, tableau {(m+n+1)}x{(2*n+m+1)} = {(m+n+1)*(2*n+m+1)*8/1024:.0f} KB", flush=True)
