"""Probe: cold config-4 level search (every node re-solved from the slack basis) -- nodes/s and a digest of the node log / node z / pivots,
to compare the register-resident group kernel (default) with LPX_RESIDENT_REGS=0 (rows in LDS) and LPX_RESIDENT_GROUP=0 (streaming)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
n, m = (int(x) for x in os.environ.get("PROBE_SHAPE", "512,256").split(","))
c, A, rel, b = synth.binary_ip(n, m)
p = L.LPProblem.from_arrays(0, c, A, rel, b)
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 800
for rep in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    t0 = time.perf_counter()
    r = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=conc, max_nodes=budget).Solve(p)
    dt = time.perf_counter() - t0
    h = hashlib.sha256(np.ascontiguousarray(r.NodeLog).view(np.uint8)).hexdigest()[:12] + "/" + hashlib.sha256(np.ascontiguousarray(r.NodeZ).view(np.uint8)).hexdigest()[:12]
    print(f"conc={conc} budget={budget}: {dt:.2f} s, LPs {r.LpSolves}, {r.LpSolves / dt:.0f} nodes/s, pivots {r.Stats['pivots']}, digest {h}", flush=True)
