"""Runs the rank-1 update kernel on the headline shape (raw 4096x8192 tableau, forced pivots) and on
config 2 (full solve) -- the target of the rocprofv3 --pmc passes whose results go to profiles/."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

L._lib.check(L._lib.lib().lpx_init(0))
npiv = int(sys.argv[1]) if len(sys.argv) > 1 else 60
HR, HC = 4096, 8192
hd = L.DeviceTableau.from_host(synth.raw_tableau(HR, HC))
rows, cols = synth.forced_pivot_list(HR, HC, npiv)
_, st = hd.forced_pivots(rows, cols, 0.1, use_graph=0, batch=20)
print("headline pivots", st["pivots"])
hd.close()
c, A, b = synth.dense_lp(1024, 2048)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
status, st = dt.primal_run(use_graph=0, batch=64, max_iter=600)
print("cfg2 pivots", st["pivots"], "status", status)
dt.close()
# the bench's headline workload: real primal solve of the m=4096 n=8192 LP (tableau 4097x12289, 403 MB)
c, A, b = synth.dense_lp(4096, 8192)
T, basis = synth.primal_tableau_from(c, A, b)
del A
dt = L.DeviceTableau.from_host(T, basis)
status, st = dt.primal_run(use_graph=0, batch=20, max_iter=npiv)
print("lp-level pivots", st["pivots"], "status", status)
dt.close()
# config 3: the fused revised iteration (rv_price streams A^T with nt loads, rv_upd_ftran reads + writes W once)
c3, A3, b3 = synth.dense_lp(4096, 8192)
rv = L.DeviceRevised(A3, -c3, b3)
status, st = rv.run(max_iter=npiv, batch=20, use_graph=0)
print("revised iterations", st["pivots"], "status", status)
rv.close()
del A3
# config 4's streaming group step (lpx_group_fused: update out of place beside the select of every live node): 64 copies of the root
# tableau (769 x 1281) pivoting in lock step, i.e. full liveness -- algorithmic bytes per launch = 64 x 16 x 769 x 1281
from linear_programming_solver_lpr381_amd._lib import default_opts
cb, Ab, relb, bb = synth.binary_ip(512, 256)
Tb, basb = synth.primal_tableau_from(cb, Ab, bb)
nodes = [L.DeviceTableau.from_host(Tb, basb) for _ in range(64)]
stg, ssg = L.multi_run(nodes, [False] * 64, default_opts(False, max_iter=npiv, resident=-1, batch=20), default_opts(True, resident=-1))
print("group steps", ssg[0]["pivots"], "statuses", set(stg))
for t_ in nodes:
    t_.close()
# config 5: the knapsack expansion kernel (bytes per bound: per-launch traffic / (grid / 64 jobs x 3 bounds))
pk, wk, capk = synth.knapsack(100_000)
kp = L.LPProblem(L.Sense.Max, pk.tolist(), [L.Constraint(wk.tolist(), L.Rel.LE, capk)])
rk = L.BranchAndBoundKnapsack(max_nodes=20000, concurrent_nodes=512).Solve(kp)
print("knapsack pops", rk.Nodes, "launches", rk.Stats["launches"])
