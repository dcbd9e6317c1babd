"""Times the resident primal loop (csrc/lpx_resident.hip) against the streaming kernels on dense random LPs."""
import sys
import time

import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

shapes = [(1024, 2048), (512, 1024), (256, 512), (768, 1280), (1200, 2000)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for m, n in shapes:
    c, A, b = synth.dense_lp(m, n)
    T, basis = synth.primal_tableau_from(c, A, b)
    with L.DeviceTableau.from_host(T, basis) as dt:
        dt.snapshot()
        for res in (1, -1):
            best = None
            for rep in range(3):
                dt.restore()
                try:
                    status, st = dt.primal_run(resident=res)
                except L.LpxError as e:
                    print(f"{m}x{n} resident={res}: {e}")
                    break
                rate = st["pivots"] / (st["loop_ms"] * 1e-3)
                best = max(best or 0, rate)
            if best:
                print(f"{m}x{n} tableau {T.shape[0]}x{T.shape[1]} resident={res:2d}: {st['pivots']} pivots, "
                      f"{best:,.0f} pivots/s ({1e6 / best:.2f} us/pivot), launches {st['launches']}", flush=True)
