"""Probe: the bench's config-4 legs in one process, in the bench's order (cold 1600 nodes, then warm 8000), with the host phase
timers on stderr (LPX_BNB_TIMING=1): where does the FIRST warm run of a process spend its time?"""
import sys, time, os
os.environ.setdefault("LPX_BNB_TIMING", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
cb, Ab, relb, bb = synth.binary_ip(512, 256)
pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
t0 = time.perf_counter(); r = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=64, max_nodes=1600).Solve(pb); dt = time.perf_counter() - t0
print(f"cold: {dt:.2f}s nodes/s={r.LpSolves/dt:.0f}", flush=True)
for k in range(2):
    t0 = time.perf_counter(); r = L.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=64, max_nodes=8000).Solve(pb); dt = time.perf_counter() - t0
    print(f"warm run {k}: {dt:.2f}s nodes/s={r.LpSolves/dt:.0f}", flush=True)
