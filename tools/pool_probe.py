import sys, os, time
sys.path.insert(0, os.getcwd())
import linear_programming_solver_lpr381_amd as L
L._lib.check(L._lib.lib().lpx_init(0))
t = time.perf_counter(); h0 = L.DeviceTableau(800, 1320); print("first handle %.1f ms" % ((time.perf_counter()-t)*1e3))
t = time.perf_counter(); hs = [L.DeviceTableau(800, 1320) for _ in range(32)]; dt = time.perf_counter()-t
print("32 handles %.1f ms -> %.2f ms each" % (dt*1e3, dt*1e3/32))
t = time.perf_counter()
for h in hs: h.close()
print("destroy 32: %.1f ms" % ((time.perf_counter()-t)*1e3))
