"""Probe: lpx_group_fused at FULL liveness -- K copies of the config-4 root tableau (769 x 1281) pivoting in lock step on the
streaming group path (resident = -1): us per step and the rate against 16*R*C bytes per node and step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
from linear_programming_solver_lpr381_amd._lib import default_opts
L._lib.check(L._lib.lib().lpx_init(0))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 256
prof = int(sys.argv[3]) if len(sys.argv) > 3 else 0
c, A, rel, b = synth.binary_ip(512, 256)
T, basis = synth.primal_tableau_from(c, A, b)
R, C = T.shape
ts = [L.DeviceTableau.from_host(T, basis) for _ in range(K)]
for rep in range(3):
    for t in ts:
        t.upload(T, basis)
    po = default_opts(False, max_iter=iters, resident=-1, batch=64, profile=prof)
    t0 = time.perf_counter()
    st, ss = L.multi_run(ts, [False] * K, po, default_opts(True, resident=-1))
    dt = time.perf_counter() - t0
    piv = sum(s["pivots"] for s in ss)
    steps = ss[0]["pivots"]
    ld = (C + 15) // 16 * 16
    msg = f"K={K} {R}x{C}: {dt*1e3:.1f} ms, {steps} steps, {1e6*dt/steps:.1f} us/step, {piv*16*R*ld/dt/1e12:.2f} TB/s (padded), statuses {set(st)}"
    if prof and ss[0]["update_launches"]:
        us = 1e3 * ss[0]["update_ms_sum"] / ss[0]["update_launches"]
        msg += f"; HIP events {us:.1f} us/launch over {ss[0]['update_launches']} = {K*16*R*ld/us/1e6:.2f} TB/s"
    print(msg, flush=True)
