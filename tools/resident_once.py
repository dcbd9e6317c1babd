"""Two resident solves of config 2 (lpx_resident_primal) -- the target of the rocprofv3 --pmc passes for that kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

c, A, b = synth.dense_lp(1024, 2048)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
for _ in range(2):
    dt.restore()
    status, st = dt.primal_run(resident=1)
    print("status", status, "pivots", st["pivots"], "loop_ms", st["loop_ms"], flush=True)
dt.close()
