"""Diagnostic: the primal streaming loop on mid-size tableaux (between the LDS-resident limit and the Infinity Cache size),
two-launch in place (LPX_FUSED_PIVOT=0) against the fused out-of-place launch, with the cache policy its launcher picks by
size (default policy up to 152 MiB, lpx_pivot_fused_c) and with the streaming mix forced (LPX_UPDATE_POLICY=2).  Knobs are read once per process: one child per setting.
Usage: python tools/probe_fused_mid.py [pivots]     (PROBE_SIZES=8192x8192,... overrides the list of m x n)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import json, sys
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
out = {}
import os
SIZES = [tuple(int(x) for x in s.split("x")) for s in os.environ["PROBE_SIZES"].split(",")] if os.environ.get("PROBE_SIZES") else None
for m, n in SIZES or [(128, 256), (256, 512), (512, 1024), (1024, 2048), (1536, 3072), (2048, 4096), (2560, 5120), (2816, 5632), (3072, 6144), (3328, 6656), (3584, 7168), (4096, 8192)]:
    c, A, b = synth.dense_lp(m, n)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    with L.DeviceTableau.from_host(T, basis) as dt:
        dt.snapshot()
        dt.primal_run(max_iter=200, resident=-1)
        best = 1e30
        for _ in range(2):
            dt.restore()
            status, st = dt.primal_run(max_iter=int(sys.argv[1]), resident=-1)
            best = min(best, 1e3 * st["loop_ms"] / max(st["pivots"], 1))
        out[f"{T.shape[0]}x{T.shape[1]} ({T.nbytes / 1e6:.0f} MB)"] = [round(best, 2), st["launches"], st["pivots"]]
print(json.dumps(out))
"""
piv = sys.argv[1] if len(sys.argv) > 1 else "2000"
res = {}
for tag, env in [("two-launch", {"LPX_FUSED_PIVOT": "0"}), ("fused", {"LPX_FUSED_PIVOT": "1"}),
                 ("fused-nt", {"LPX_FUSED_PIVOT": "1", "LPX_UPDATE_POLICY": "2"})]:
    r = subprocess.run([sys.executable, "-c", CHILD, piv], env=dict(os.environ, PYTHONPATH=ROOT, **env), capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
for k in res["two-launch"]:
    a, b, c = res["two-launch"][k], res["fused"][k], res["fused-nt"][k]
    print(f"{k:28s} two-launch {a[0]:8.2f} us/pivot   fused (policy by size) {b[0]:8.2f} x{a[0] / b[0]:.3f}   fused, streaming mix forced {c[0]:8.2f} x{a[0] / c[0]:.3f}")
