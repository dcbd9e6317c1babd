import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
c, A, rel, b = synth.binary_ip(512, 256)
T, basis = synth.primal_tableau_from(c, A, b)
with L.DeviceTableau.from_host(T, basis) as dt:
    dt.snapshot()
    for res in (1, -1):
        for rep in range(2):
            dt.restore()
            status, st = dt.primal_run(resident=res)
        print(f"config-4 root LP {T.shape} resident={res}: status={status} pivots={st['pivots']} {1e3*st['loop_ms']/st['pivots']:.2f} us/pivot")
    # the same through the group kernel (dual entry point on a primal-feasible tableau: FDF does the work)
    for rep in range(2):
        dt.restore()
        status, st = dt.dual_run(L.default_opts(True, fdf_guard=100000, cleanup=1, resident=1))
    print(f"  group kernel (dual entry): status={status} pivots={st['pivots']} fdf={st['fdf_pivots']} {1e3*st['loop_ms']/st['pivots']:.2f} us/pivot")
