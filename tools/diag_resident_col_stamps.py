"""Diagnostic (not part of the product): cycles per phase of the column-owning resident primal loop, workgroup 1 lane 0.
Needs the -DLPX_STAMPS build (make -C linear_programming_solver_lpr381_amd/csrc stamps):
  LPX_LIB_PATH=.../csrc/build/liblpx_stamps.so python tools/diag_resident_col_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

lib = L._lib.lib()
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
c, A, b = synth.dense_lp(m, n)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
dt.primal_run(resident=1)
dt.restore()
out = (C.c_ulonglong * 16)()
lib.lpx_debug_resident_col(dt._h, out, 16, 1)
status, st = dt.primal_run(resident=1)
lib.lpx_debug_resident_col(dt._h, out, 16, 0)
v = list(out)
piv = st["pivots"]
names = ["gather candidates (exchange)", "winner reduction", "gather winner's column", "ratios + hysteresis scan",
         "pivot row slice + next candidate", "candidate column update + publish", "round-end barrier (wait for the other waves)",
         "bulk update (this wave)"]
tot = sum(v[:8])
us = 1e3 * st["loop_ms"] / piv
print(f"{m}x{n}: pivots={piv} us/pivot={us:.2f}; s_memtime ticks/pivot={tot / piv:.0f}")
for nm, x in zip(names, v[:8]):
    print(f"  {nm:36s} {x / piv:8.1f} ticks  {100 * x / tot:5.1f}%  ~{x / tot * us:.2f} us")
