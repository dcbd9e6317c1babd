"""Diagnostic (not part of the product): cycles per phase of the resident group kernel on one config-4 node LP while K copies of it run
beside it in the same launch (contention between nodes).  Needs the -DLPX_STAMPS build:
  LPX_LIB_PATH=.../csrc/build/liblpx_stamps.so python tools/diag_group_stamps_loaded.py [K ...]"""
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
names = ["lookahead publish (+barrier)", "gather (ratio, rhs) (exchange 1)", "decision (hysteresis / leaving row)",
         "pivot row (exchange 2)", "factors + objective update + argmin", "rank-1 update + barrier"]
o = L.default_opts(True, fdf_guard=10000, cleanup=1)
po = L.default_opts(False)
for K in [int(a) for a in sys.argv[1:]] or [1, 4, 8, 12]:
    hs = [L.DeviceTableau.from_host(T, basis) for _ in range(K)]
    L.multi_run(hs, [1] * K, po, o)                    # warm-up: allocations, exchange buffers
    for h in hs: h.upload(T, basis)
    out = (C.c_ulonglong * 16)()
    lib.lpx_debug_resident_group(hs[0]._h, out, 16, 1)
    st, stats = L.multi_run(hs, [1] * K, po, o)
    lib.lpx_debug_resident_group(hs[0]._h, out, 16, 0)
    v = list(out); piv = stats[0]["pivots"]; tot = sum(v[:6])
    us = 1e3 * stats[0]["loop_ms"] / piv
    print(f"{K} nodes in the launch: status={st[0]} pivots={piv} us/pivot (whole call)={us:.2f}; ticks/pivot={tot / piv:.0f}")
    for nm, x in zip(names, v[:6]):
        print(f"  {nm:38s} {x / piv:8.1f} ticks  {100 * x / tot:5.1f}%")
    for h in hs: h.close()
