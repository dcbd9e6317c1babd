"""Probe: wall-clock cost of one resident launch that cannot finish (LPX_RESIDENT_TEST_MUTE=1) + the hand-over."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
c, A, b = synth.dense_lp(1024, 2048)
T, basis = synth.primal_tableau_from(c, A, b)
dt = L.DeviceTableau.from_host(T, basis)
t0 = time.perf_counter(); status, st = dt.primal_run(); t1 = time.perf_counter() - t0
print(f"MUTE={os.environ.get('LPX_RESIDENT_TEST_MUTE','0')}: first solve {t1:.3f} s (pivots {st['pivots']}, launches {st['launches']}, loop_ms {st['loop_ms']:.1f})")
dt.upload(T, basis)
t0 = time.perf_counter(); status, st = dt.primal_run(); t1 = time.perf_counter() - t0
print(f"  second solve on the same handle {t1:.3f} s (launches {st['launches']})")
