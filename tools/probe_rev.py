"""Probe: config 3 (revised simplex m=4096 n=8192), iterations/s of the device loop."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8192)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 300
c, A, b = synth.dense_lp(m, n)
rv = L.DeviceRevised(A, -c, b); rv.run(max_iter=20, batch=20); rv.close()
rv = L.DeviceRevised(A, -c, b)
st, s = rv.run(max_iter=iters, batch=50)
print(f"m={m} n={n}: status={st} iterations={s['pivots']} loop_ms={s['loop_ms']:.2f} us/iter={1e3*s['loop_ms']/max(s['pivots'],1):.2f}", flush=True)
print("residual", rv.residual())
rv.set_refactor_mode(1)
t0 = time.perf_counter(); rv.refactor(); print(f"fast refactor (first: allocations) {time.perf_counter()-t0:.4f} s", rv.refactor_stats(), rv.residual())
st, s = rv.run(max_iter=iters, batch=50)
t0 = time.perf_counter(); rv.refactor(); print(f"fast refactor {time.perf_counter()-t0:.4f} s", rv.refactor_stats(), rv.residual())
rv.set_refactor_mode(0)
t0 = time.perf_counter(); rv.refactor(); print(f"exact refactor {time.perf_counter()-t0:.3f} s", rv.residual())
rv.close()
