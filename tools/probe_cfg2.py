"""Probe: config 2 full solve on the resident kernels (column-owning vs row-owning workgroups) and the streaming kernels."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
shapes = [(1024, 2048), (256, 512), (768, 512)]
for (m, n) in shapes:
    c, A, b = synth.dense_lp(m, n)
    T, basis = synth.primal_tableau_from(c, A, b)
    dt = L.DeviceTableau.from_host(T, basis); dt.snapshot()
    for rep in range(2):
        dt.restore()
        t0 = time.perf_counter(); status, st = dt.primal_run(); t1 = time.perf_counter() - t0
    print(f"m={m} n={n} LPX_RESIDENT_COL={os.environ.get('LPX_RESIDENT_COL','1')}: status {status} pivots {st['pivots']} launches {st['launches']} "
          f"{t1*1e3:.2f} ms  {1e6*t1/max(st['pivots'],1):.2f} us/pivot  {st['pivots']/t1:.0f} pivots/s", flush=True)
    dt.close()
