"""lpx_update_mb on the two headline shapes (forced pivots, HIP-event kernel time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
for (R, C) in [(4096, 8192), (4097, 12289), (1025, 3073)]:
    hd = L.DeviceTableau.from_host(synth.raw_tableau(R, C))
    rows, cols = synth.forced_pivot_list(R, C, 110)
    hd.forced_pivots(rows[:10], cols[:10], 0.1)
    _, st = hd.forced_pivots(rows[10:], cols[10:], 0.1, profile=1, batch=100)
    ms = st["update_ms_sum"] / st["update_launches"]
    print(f"{R}x{C}: {1e3*ms:.1f} us -> {16.0*R*C/(ms*1e-3)/1e12:.2f} TB/s", flush=True)
    hd.close()
