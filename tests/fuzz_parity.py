"""Randomized parity sweep (GPU vs CPU oracle, bitwise): many shapes and seeds, dense / integer / degenerate /
near-tie data, primal and dual loops, single runs and batched groups.  Prints a summary; exits 1 on mismatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth
from oracle import oracle as O

L._lib.check(L._lib.lib().lpx_init(0))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 12345)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def make(kind, m, n):
    if kind == 0:      # dense uniform
        A = rng.random((m, n)); b = 0.5 * n * rng.uniform(0.9, 1.1, m); c = rng.uniform(0.5, 1.5, n)
    elif kind == 1:    # small integers: many exact ties, zero ratios
        A = rng.integers(0, 4, (m, n)).astype(float); b = rng.integers(0, 3, m).astype(float); c = rng.integers(1, 6, n).astype(float)
    elif kind == 2:    # near ties inside the hysteresis band
        A = np.ones((m, n)) + 1e-10 * rng.integers(0, 20, (m, n)); b = 5.0 + 1e-10 * rng.integers(0, 30, m); c = 1.0 + 1e-10 * rng.integers(0, 9, n)
    else:              # sparse-ish with negative entries
        A = rng.normal(size=(m, n)) * (rng.random((m, n)) < 0.4); b = rng.uniform(0.0, 3.0, m); c = rng.normal(size=n)
    return c, A, b


t0 = time.time()
bad = 0
checked = 0
group, gref, gdual = [], [], []
for it in range(N):
    kind = int(rng.integers(0, 4))
    m = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 100, 257, 700, 1030, 1500]))
    n = int(rng.choice([1, 2, 5, 11, 32, 70, 129, 300]))
    c, A, b = make(kind, m, n)
    T, basis = synth.primal_tableau_from(c, A, b)
    dual = bool(rng.integers(0, 2)) and m >= 2
    if dual:
        k = int(rng.integers(1, max(2, m // 3)))
        for i in rng.choice(m, size=k, replace=False):
            T[i, :n] *= -1.0
            T[i, -1] = -abs(T[i, -1]) * 0.05
    cap = 400
    Tr, br = T.copy(), basis.copy()
    if dual:
        cleanup = int(rng.integers(0, 2)); guard = int(rng.choice([0, 3, 100, cap]))
        st_ref, tr_ref, _ = O.dual_tableau(Tr, br, fdf_guard=guard, max_iter=cap, cleanup=cleanup)
    else:
        st_ref, tr_ref = O.primal_tableau(Tr, br, max_iter=cap)
    if rng.random() < 0.35 and (not dual or (guard == cap and cleanup == 1)):
        group.append(L.DeviceTableau.from_host(T, basis)); gref.append((st_ref, tr_ref, Tr, br)); gdual.append(dual)
        if len(group) == 7:
            sts, _ = L.multi_run(group, gdual, L.default_opts(False, max_iter=cap), L.default_opts(True, fdf_guard=cap, cleanup=1, max_iter=cap))
            for t, (s0, tr0, T0, b0), s1 in zip(group, gref, sts):
                Tg, bg = t.download(); trg = t.trace(); t.close(); checked += 1
                if s1 != s0 or trg.tolist() != tr0.tolist() or not np.array_equal(bits(Tg), bits(T0)) or bg.tolist() != b0.tolist():
                    bad += 1; print("MISMATCH (group)", T0.shape, s1, s0, len(trg), len(tr0))
            group, gref, gdual = [], [], []
        continue
    with L.DeviceTableau.from_host(T, basis) as dt:
        batch = int(rng.choice([1, 7, 64])); graph = int(rng.integers(0, 2))
        if dual:
            st, _ = dt.dual_run(fdf_guard=guard, max_iter=cap, cleanup=cleanup, batch=batch, use_graph=graph)
        else:
            st, _ = dt.primal_run(max_iter=cap, batch=batch, use_graph=graph)
        Tg, bg = dt.download(); trg = dt.trace()
    checked += 1
    if st != st_ref or trg.tolist() != tr_ref.tolist() or not np.array_equal(bits(Tg), bits(Tr)) or bg.tolist() != br.tolist():
        bad += 1
        print("MISMATCH", "dual" if dual else "primal", "kind", kind, T.shape, "status", st, st_ref, "pivots", len(trg), len(tr_ref))
for t in group:
    t.close()
print(f"fuzz: {checked} instances checked, {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
