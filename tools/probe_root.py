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
    # three (or six) copies of the same node through lpx_multi_run on the group kernel: the cost of sharing the chip
os.environ.setdefault("LPX_RESIDENT_GROUP", "1")
import time
for ncopy in (1, 3, 6):
    hs = [L.DeviceTableau.from_host(T, basis) for _ in range(ncopy)]
    o = L.default_opts(True, fdf_guard=100000, cleanup=1)
    L.multi_run(hs, [1] * ncopy, dual_opts=o)
    for h in hs: h.upload(T, basis)
    t0 = time.perf_counter()
    sts, stats = L.multi_run(hs, [1] * ncopy, dual_opts=o)
    dt_ = time.perf_counter() - t0
    piv = sum(s["pivots"] for s in stats)
    print(f"  multi_run x{ncopy}: {piv} pivots in {dt_*1e3:.2f} ms -> {piv/dt_:,.0f} pivots/s aggregate ({dt_*1e6*min(ncopy,3)/piv:.2f} us per pivot per resident node), launches {stats[0]['launches']}")
    for h in hs: h.close()
