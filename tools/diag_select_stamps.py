"""Diagnostic (not part of the product): where does lpx_select_la spend its cycles?
Needs the -DLPX_STAMPS build:  LPX_LIB_PATH=.../csrc/build/liblpx_stamps.so python tools/diag_select_stamps.py
"""
import ctypes as C
import os
import sys

os.environ.setdefault("LPX_FUSED_PIVOT", "0")     # the stamps live in lpx_select_la / lpx_select_mb: the two-launch path

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth

lib = L._lib.lib()
L._lib.check(lib.lpx_init(0))
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
c, A, b = synth.dense_lp(m, n)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
dt.snapshot()
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
dt.primal_run(max_iter=cap)
dt.restore()
out = (C.c_ulonglong * 16)()
lib.lpx_debug_ws(dt._h, out, 16, 1)
hs = (C.c_ulonglong * 32)()
lib.lpx_debug_hs(hs, 1)

status, st = dt.primal_run(use_graph=1, batch=64, max_iter=cap)
lib.lpx_debug_ws(dt._h, out, 16, 0)
lib.lpx_debug_hs(hs, 0)
v = list(out)
calls = v[15]
names = ["state load+branch", "ratio test (hysteresis)", "row normalise + lookahead loop", "block argmin", "tail stores"]
tot = sum(v[:5])
print(f"pivots={st['pivots']} calls={calls} loop_ms={st['loop_ms']:.2f}  us/pivot={1e3*st['loop_ms']/st['pivots']:.2f}")
clk = tot / (v[14] / 100e6) / 1e9 if v[14] else 0
print(f"in-kernel clock ~{clk:.2f} GHz; total stamped {tot/calls:.0f} cycles/call = {tot/calls/clk/1e3:.2f} us")
for nm, x in zip(names, v[:5]):
    print(f"  {nm:34s} {x/calls:9.0f} cycles  {100*x/tot:5.1f}%")

hn = ["operand loads (32/lane)", "16 divisions", "local min + DPP reduce", "band check (fast path taken)", "band check (slow path)", "ballot chain"]
print("ratio scan breakdown (workgroup 1, wave 0; the lpx_g_stamps symbol is per code object):")
for nm, x in zip(hn, list(hs)[:6]):
    print(f"  {nm:34s} {x/calls:9.0f} cycles")
