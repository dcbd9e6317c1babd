"""Calibration probe (not part of the product): rough timings of configs 3-5 on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

L._lib.check(L._lib.lib().lpx_init(0))
which = sys.argv[1:] or ["rev", "bnb", "knap"]
if "rev" in which:
    for (m, n, it) in [(1024, 2048, 300), (4096, 8192, 200)]:
        c, A, b = synth.dense_lp(m, n)
        t = time.perf_counter(); rv = L.DeviceRevised(A, -c, b); t_up = time.perf_counter() - t
        st, s = rv.run(max_iter=it, batch=50)
        print(f"revised m={m} n={n}: status={st} iters={s['pivots']} loop_ms={s['loop_ms']:.1f} -> {s['pivots']/(s['loop_ms']*1e-3):.0f} it/s (upload {t_up:.2f}s)")
        rv.close()
if "bnb" in which or "wide" in which:
    c, A, rel, b = synth.binary_ip(512, 256)
    p = L.LPProblem.from_arrays(0, c, A, rel, b)
    for (mode, search, cn, mx) in ([(0, 0, 1, 16), (1, 0, 1, 12), (1, 1, 4, 24), (1, 1, 8, 48)] if 'wide' not in which else [(1, 2, 64, 1600)] if 'dive' not in which and 'cold' not in which else [(1, 1, 32, 400), (1, 1, 64, 400), (1, 1, 12, 400)] if 'cold' in which else [(1, 2, 64, 20000), (1, 1, 32, 1500)]):
        t = time.perf_counter()
        r = L.BranchAndBound(bnb_mode=mode, bnb_search=search, concurrent_nodes=cn, max_nodes=mx, bnb_dive=1 if "dive" in which else 0).Solve(p)
        dt = time.perf_counter() - t
        print(f"bnb mode={mode} search={search} conc={cn} depth={int(r.NodeLog[:,0].max()) if len(r.NodeLog) else 0}: nodes={r.Nodes} lp_solves={r.LpSolves} pivots={r.Stats['pivots']} best={r.OptimalValue} {dt:.2f}s -> {r.LpSolves/dt:.1f} LP/s, {r.Stats['pivots']/dt:.0f} pivots/s")
if "knap" in which:
    p, w, cap = synth.knapsack(100000)
    kp = L.LPProblem(L.Sense.Max, p.tolist(), [L.Constraint(w.tolist(), L.Rel.LE, cap)])
    for spec in (1, 64, 512):
        t = time.perf_counter()
        r = L.BranchAndBoundKnapsack(max_nodes=20000, concurrent_nodes=spec).Solve(kp)
        dt = time.perf_counter() - t
        print(f"knapsack n=100000 spec={spec}: popped={r.Nodes} relax={r.Aux[0]:.0f} launches={r.Stats['launches']} best={r.OptimalValue} {dt:.2f}s -> {r.Nodes/dt:.0f} nodes/s")
