"""Randomized sweep of lpx_multi_run: groups of 1 ... 400 node LPs (more than the chip has compute units), small and mid-size shapes mixed
in one group, primal and dual loops mixed, every result against the CPU oracle bit for bit.  python tests/fuzz_groups.py [seed] [trials]"""
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
from oracle import oracle as O
L._lib.check(L._lib.lib().lpx_init(0))
def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    count = int(rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 400]))
    big = trial % 4 == 3
    hs, want = [], []
    for k in range(count):
        if big: m, n = int(rng.integers(200, 420)), int(rng.integers(300, 900))
        else: m, n = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        c, A, b = synth.dense_lp(m, n, seed=int(rng.integers(1, 1 << 30)))
        T, basis = synth.primal_tableau_from(c, A, b)
        dual = int(rng.integers(0, 2)) if m > 1 else 0
        if dual:
            i = int(rng.integers(0, m)); T[i, :n] *= -1.0; T[i, -1] = -0.02 * T[i, -1]
        Tr, br = T.copy(), basis.copy()
        if dual: st, tr, _ = O.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
        else: st, tr = O.primal_tableau(Tr, br)
        want.append((st, len(tr), Tr, br, dual)); hs.append(L.DeviceTableau.from_host(T, basis))
        if big and k >= 40: break
    st, stats = L.multi_run(hs, [w[4] for w in want], None, L.default_opts(True, fdf_guard=10000, cleanup=1))
    ok = True
    for dt, s, k, (ws, wp, Tr, br, dual) in zip(hs, st, stats, want):
        Tg, bg = dt.download()
        if not (s == ws and k["pivots"] == wp and bg.tolist() == br.tolist() and np.array_equal(bits(Tg), bits(Tr))): ok = False
        dt.close()
    print(f"trial {trial}: {len(hs)} LPs {'big' if big else 'small'} mixed primal/dual: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
