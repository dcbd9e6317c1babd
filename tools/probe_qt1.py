"""Probe for the record gpurun_out/r2_qt1.log: two primal runs of one handle under `rocprofv3 --kernel-trace`, the second
with another iteration cap (another graph key).  Prints the executable mappings first so a raw stack trace can be attributed.
QT1_CAPS=2000,10000 (default) / 2000,2000 (same key: no second graph)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
m, n = (int(x) for x in os.environ.get("QT1_SHAPE", "4096,8192").split(","))
c, A, b = synth.dense_lp(m, n)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
for line in open("/proc/self/maps"):
    if "r-xp" in line:
        print(line.rstrip())
sys.stdout.flush()
for mi in (int(x) for x in os.environ.get("QT1_CAPS", "2000,10000").split(",")):
    dt.restore()
    status, st = dt.primal_run(max_iter=mi, resident=-1)
    print(f"max_iter={mi}: status={status} pivots={st['pivots']} launches={st['launches']}", flush=True)
