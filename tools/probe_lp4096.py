"""Probe: how many pivots does the m=4096 n=8192 LP (bench headline workload) need, and at what rate?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
m, n = 4096, 8192
c, A, b = synth.dense_lp(m, n)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
for mi in (2000, 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 200000):
    dt.restore()
    t0 = time.perf_counter()
    status, st = dt.primal_run(max_iter=mi)
    dtm = time.perf_counter() - t0
    print(f"max_iter={mi}: status={status} pivots={st['pivots']} wall={dtm:.3f}s loop_ms={st['loop_ms']:.1f} us/pivot={1e3*st['loop_ms']/max(st['pivots'],1):.2f}", flush=True)
