"""Diagnostic (not part of the product): cycles per phase of lpx_resident_group on one config-4 node LP (dual path).
Needs the -DLPX_STAMPS build:  LPX_LIB_PATH=.../csrc/build/liblpx_stamps.so python tools/diag_group_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

lib = L._lib.lib()
c, A, rel, b = synth.binary_ip(512, 256)
n = len(c)
A2 = np.vstack([A, -np.eye(n)[:1]]); b2 = np.concatenate([b, [-1.0]])        # x_1 >= 1 as -x_1 <= -1 (repaired mode row)
T, basis = synth.primal_tableau_from(c, A2, b2)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
o = L.default_opts(True, fdf_guard=10000, cleanup=1, resident=1)
dt.dual_run(o)
dt.restore()
out = (C.c_ulonglong * 16)()
lib.lpx_debug_resident_group(dt._h, out, 16, 1)
status, st = dt.dual_run(o)
lib.lpx_debug_resident_group(dt._h, out, 16, 0)
v = list(out)
piv = st["pivots"]
names = ["lookahead publish (+barrier)", "gather (ratio, rhs) (exchange 1)", "decision (hysteresis / leaving row)",
         "pivot row (exchange 2)", "factors + objective update + argmin", "rank-1 update (LDS) + barrier"]
tot = sum(v[:6])
us = 1e3 * st["loop_ms"] / piv
print(f"node LP {T.shape}: status={status} pivots={piv} (fdf {st['fdf_pivots']}) us/pivot={us:.2f}; ticks/pivot={tot / piv:.0f}")
for nm, x in zip(names, v[:6]):
    print(f"  {nm:38s} {x / piv:8.1f} ticks  {100 * x / tot:5.1f}%  ~{x / tot * us:.2f} us")
