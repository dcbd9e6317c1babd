"""Times one dual-simplex node LP of config 4 (x_1 >= 1 child) on the resident group kernel vs the streaming kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

c, A, rel, b = synth.binary_ip(512, 256)
n = len(c)
A2 = np.vstack([A, np.eye(n)[:1]]); rel2 = np.concatenate([rel, [1]]); b2 = np.concatenate([b, [1.0]])
p = L.LPProblem.from_arrays(0, c, A2, rel2, b2)
for rep in range(3):
    t = time.perf_counter()
    r = L.DualSimplex(dual_flags=7).Solve(p)
    dt = time.perf_counter() - t
    s = r.Stats
    print(f"dual node LP: status={r.Status} pivots={s['pivots']} (fdf {s['fdf_pivots']}, cleanup {s['cleanup_pivots']}) launches={s['launches']} "
          f"loop_ms={s['loop_ms']:.2f} -> {1e3 * s['loop_ms'] / max(s['pivots'], 1):.2f} us/pivot; solve wall {dt * 1e3:.1f} ms", flush=True)
