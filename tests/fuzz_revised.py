"""Randomized sweep of the revised path: ragged shapes (m = 1 ... 700, n below and above m, widths that are no multiple of the kernels'
128-column chunks), every pivot sequence against the CPU oracle, z and x_B within 1e-9.  python tests/fuzz_revised.py [seed] [trials]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
from oracle import oracle as O
L._lib.check(L._lib.lib().lpx_init(0))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 24):
    if trial % 6 == 5: m, n = int(rng.integers(300, 700)), int(rng.integers(200, 900))
    else: m, n = int(rng.integers(1, 140)), int(rng.integers(1, 200))
    c, A, b = synth.dense_lp(m, n, seed=int(rng.integers(1, 1 << 30)))
    p = O.Problem(O.MAX, c, A, np.zeros(m, np.int32), b)
    ref = O.revised_solve(p)
    with L.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run()
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
    ok = (status == ref.status and tr.tolist() == ref.trace.tolist() and Bidx.tolist() == ref.Bidx.tolist() and Nidx.tolist() == ref.Nidx.tolist())
    if ok and ref.status == 0:
        ok = abs(z - ref.z_internal) <= 1e-9 * max(1.0, abs(ref.z_internal)) and np.allclose(xB, ref.xB, rtol=1e-9, atol=1e-9)
    print(f"trial {trial}: m={m} n={n} status={status} pivots={len(tr)}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
