import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
L._lib.check(L._lib.lib().lpx_init(0))
c, A, rel, b = synth.binary_ip(512, 256)
n = len(c)
A2 = np.vstack([A, -np.eye(n)[:1]]); b2 = np.concatenate([b, [-1.0]])
T, basis = synth.primal_tableau_from(c, A2, b2)
o = L.default_opts(True, fdf_guard=10000, cleanup=1); po = L.default_opts(False)
# final (optimal) tableau of the node: a second run needs 0 pivots
d0 = L.DeviceTableau.from_host(T, basis); L.multi_run([d0], [1], po, o); Tf, bf = d0.download(); d0.close()
for K in (12, 96, 384):
    hs = [L.DeviceTableau.from_host(Tf, bf) for _ in range(K)]
    L.multi_run(hs, [1] * K, po, o)
    for h in hs: h.upload(Tf, bf)
    t0 = time.perf_counter(); st, stats = L.multi_run(hs, [1] * K, po, o); dt = time.perf_counter() - t0
    print(f"{K} finished nodes: {1e3*dt:.2f} ms whole call, pivots {sum(s['pivots'] for s in stats)}, per node-slot {1e6*dt/(K/12):.0f} us (12 slots)", flush=True)
    for h in hs: h.upload(T, basis)
    t0 = time.perf_counter(); st, stats = L.multi_run(hs, [1] * K, po, o); dt = time.perf_counter() - t0
    piv = sum(s['pivots'] for s in stats)
    print(f"{K} full nodes: {1e3*dt:.2f} ms, pivots {piv}, per node-slot {1e6*dt/(K/12):.0f} us = {piv/K:.0f} pivots x {1e6*dt/(K/12)/(piv/K):.1f} us", flush=True)
    for h in hs: h.close()
