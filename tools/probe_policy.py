"""Probe: the three forms of the rank-1 update (LPX_UPDATE_POLICY = 0 cache-resident / 1 all non-temporal / 2 one row in three
stored with the default policy) on one R x C tableau, forced pivots in profile mode (HIP events around every update launch).
Usage: probe_policy.py R C [pivots]   -- one process per policy (the knob is read once)."""
import os, subprocess, sys
R, C = int(sys.argv[1]), int(sys.argv[2])
npiv = int(sys.argv[3]) if len(sys.argv) > 3 else 30
if len(sys.argv) > 4:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import linear_programming_solver_lpr381_amd as L
    from linear_programming_solver_lpr381_amd import synth
    L._lib.check(L._lib.lib().lpx_init(0))
    dt = L.DeviceTableau.from_host(synth.raw_tableau(R, C))
    rows, cols = synth.forced_pivot_list(R, C, npiv)
    _, st = dt.forced_pivots(rows, cols, 0.1, use_graph=0, batch=10, profile=1)
    us = 1e3 * st["update_ms_sum"] / max(st["update_launches"], 1)
    print(f"policy {os.environ.get('LPX_UPDATE_POLICY')}: {R}x{C} ({8*R*C/1e6:.0f} MB) update {us:.1f} us = {16.0*R*C/us/1e3:.0f} GB/s "
          f"({16.0*R*C/us/1e3/8000:.3f} of 8 TB/s), {st['update_launches']} launches", flush=True)
else:
    for pol in ("1", "2"):
        env = dict(os.environ, LPX_UPDATE_POLICY=pol)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(R), str(C), str(npiv), "child"], env=env, check=True)
