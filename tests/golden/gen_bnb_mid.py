"""Generates tests/golden/bnb_mid.json: the reference-order DFS of the ORACLE (oracle/bnb.c, repaired mode) on the mid-size 0/1 IPs the
sharded searches are tested / benchmarked on, plus an independent optimum from SciPy's HiGHS.  Minutes of CPU: run once, commit the output."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from scipy.optimize import milp, LinearConstraint, Bounds
from linear_programming_solver_lpr381_amd import synth
from oracle import oracle as O

out = {}
for n, m in [(96, 24), (128, 32)]:
    c, A, rel, b = synth.binary_ip(n, m)
    res = milp(-c, constraints=LinearConstraint(A[:m], -np.inf, b[:m]), integrality=np.ones(n), bounds=Bounds(0, 1))
    entry = {"n": n, "m": m, "seed": synth.SEED, "highs_z": float(-res.fun)}
    if (n, m) == (96, 24):
        t0 = time.perf_counter()
        r = O.bnb_solve(O.Problem(O.MAX, c, A, rel.astype(np.int32), b), 1)
        entry.update({"dfs_z": float(r.best_z), "dfs_x": [int(v) for v in r.best_x], "dfs_lp_solves": int(r.lp_solves), "dfs_seconds": round(time.perf_counter() - t0, 1)})
    out[f"{n}x{m}"] = entry
    print(entry, flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "bnb_mid.json"), "w"), indent=1)
