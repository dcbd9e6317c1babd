"""GPU parity at sizes that exercise the multi-segment / ragged / full-size code paths:
ratio scans longer than one 64x16 register segment (m > 1024), odd and tiny shapes, deep knapsack
nodes (fixed lists beyond the register cache), and BASELINE.json's config 2 at full size."""
import hashlib

import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("m,n,seed", [(1025, 24, 1), (1100, 60, 2), (2500, 40, 3), (4099, 16, 4)])
def test_primal_more_rows_than_one_scan_segment(gpu, oracle, m, n, seed):
    c, A, b = synth.dense_lp(m, n, seed=seed)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.primal_run()
        Tg, bg = dt.download()
        tr = dt.trace()
    assert status == st_ref and tr.tolist() == tr_ref.tolist() and len(tr) > 0
    assert np.array_equal(_bits(Tg), _bits(Tr)) and bg.tolist() == br.tolist()


def test_degenerate_ties_across_segments(gpu, oracle):
    """0/1-style degeneracy: many exact zero ratios spread over several scan segments, plus
    near-ties inside the 1e-9 band -- the cases that leave the fast path of the hysteresis scan."""
    m, n = 2300, 6
    g = np.random.default_rng(5)
    A = g.integers(0, 3, size=(m, n)).astype(float)
    b = g.integers(0, 2, size=m).astype(float)          # about half the rows have rhs 0
    b[[7, 1500, 2299]] += np.array([3e-10, 6e-10, 9e-10])
    c = g.integers(1, 9, size=n).astype(float)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, _ = dt.primal_run()
        Tg, bg = dt.download()
        assert dt.trace().tolist() == tr_ref.tolist()
    assert status == st_ref and np.array_equal(_bits(Tg), _bits(Tr))


@pytest.mark.parametrize("m,n,seed,n_ge", [(1100, 30, 3, 40), (1300, 2100, 4, 25)])
def test_dual_long_rows_and_columns(gpu, oracle, m, n, seed, n_ge):
    c, A, b = synth.dense_lp(m, n, seed=seed)
    T, basis = synth.primal_tableau_from(c, A, b)
    g = np.random.Generator(np.random.PCG64(seed))
    for i in g.choice(m, size=n_ge, replace=False):
        T[i, :n] *= -1.0
        T[i, -1] = -0.02 * T[i, -1]
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref, nf = oracle.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1, max_iter=300)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.dual_run(fdf_guard=10000, cleanup=1, max_iter=300)
        Tg, bg = dt.download()
        assert dt.trace().tolist() == tr_ref.tolist()
    assert status == st_ref and np.array_equal(_bits(Tg), _bits(Tr))


def test_revised_more_rows_than_one_segment(gpu, oracle):
    m, n = 1100, 40
    c, A, b = synth.dense_lp(m, n, seed=6)
    # the oracle re-inverts an m x m basis every iteration: keep the run short
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b), max_iter=6)
    with gpu.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run(max_iter=6)
        assert rv.trace().tolist() == ref.trace.tolist()
        Bidx, Nidx, xB, z = rv.result()
    assert Bidx.tolist() == ref.Bidx.tolist()
    assert abs(z - ref.z_internal) <= 1e-9 * abs(ref.z_internal)


@pytest.mark.parametrize("R,C", [(2, 3), (2, 130), (3, 17), (9, 1), (70, 2)])
def test_tiny_and_skinny_shapes(gpu, oracle, R, C):
    if C < 2:
        with pytest.raises(gpu.LpxError):
            gpu.DeviceTableau(R, C)
        return
    g = np.random.default_rng(R * 100 + C)
    T = g.uniform(-1, 1, size=(R, C))
    T[:-1, -1] = np.abs(T[:-1, -1])
    basis = np.arange(R - 1, dtype=np.int32)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br, max_iter=50)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, _ = dt.primal_run(max_iter=50)
        Tg, bg = dt.download()
        assert dt.trace().tolist() == tr_ref.tolist()
    assert status == st_ref and np.array_equal(_bits(Tg), _bits(Tr), equal_nan=False) or np.array_equal(_bits(Tg), _bits(Tr))


def test_set_shape_reuses_one_handle(gpu, oracle):
    """A capacity-sized handle takes smaller tableaux (B&B depths) without re-creating anything."""
    import ctypes as C
    cap = gpu.DeviceTableau(200, 330)
    for (m, n, seed) in [(40, 60, 2), (64, 100, 3), (8, 12, 1)]:
        c, A, b = synth.dense_lp(m, n, seed=seed)
        T, basis = synth.primal_tableau_from(c, A, b)
        gpu._lib.check(gpu._lib.lib().lpx_tableau_set_shape(cap._h, T.shape[0], T.shape[1]))
        cap.R, cap.C = T.shape
        cap.upload(T, basis)
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = oracle.primal_tableau(Tr, br)
        status, _ = cap.primal_run()
        Tg, bg = cap.download()
        assert status == st_ref and cap.trace().tolist() == tr_ref.tolist()
        assert np.array_equal(_bits(Tg), _bits(Tr))
    with pytest.raises(gpu.LpxError):
        gpu._lib.check(gpu._lib.lib().lpx_tableau_set_shape(cap._h, 201, 330))
    cap.close()


def test_knapsack_deep_nodes_beyond_register_cache(gpu, oracle):
    g = np.random.default_rng(9)
    n = 3000
    w = g.integers(1, 1001, size=n).astype(float)
    p = w + g.integers(0, 101, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    dk = gpu.DeviceKnapsack(p, w, cap)
    order = oracle.knapsack_order(p, w)
    nodes = []
    for depth in (255, 256, 257, 300, 700, 1500):
        idx = g.choice(n, size=depth, replace=False)
        nodes.append({int(i): int(g.integers(0, 4) == 0) for i in idx})   # mostly fixed out: stays feasible
    P, W, F, X = dk.relax_batch(nodes)
    for j, nd in enumerate(nodes):
        a = -np.ones(n, np.int32)
        for i, v in nd.items():
            a[i] = v
        rp, rw, rf, rx = oracle.knapsack_relax(p, w, cap, order, a, want_vector=True)
        assert (P[j], W[j], F[j]) == (rp, rw, rf), (j, len(nd))
    dk.close()


def test_config2_full_size_bitwise(gpu, oracle):
    """BASELINE.json config 2 (m=1024, n=2048) at full size: 8972 pivots, every one identical."""
    c, A, b = synth.dense_lp(1024, 2048)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.primal_run()
        Tg, bg = dt.download()
        tr = dt.trace()
    assert status == st_ref == 0 and st["pivots"] == len(tr_ref) == 8972
    assert hashlib.sha256(tr.tobytes()).hexdigest() == hashlib.sha256(tr_ref.tobytes()).hexdigest()
    assert np.array_equal(_bits(Tg), _bits(Tr)) and bg.tolist() == br.tolist()


def test_headline_shape_properties(gpu):
    """4096x8192 (north-star shape): size-independent properties of Gauss-Jordan pivots -- after a
    pivot at (r,q) column q is exactly e_r, row r is scaled by 1/piv, and pivoting is idempotent on
    that column."""
    R, C = 4096, 8192
    T0 = synth.raw_tableau(R, C)
    rows, cols = synth.forced_pivot_list(R, C, 6)
    with gpu.DeviceTableau.from_host(T0) as dt:
        chosen, st = dt.forced_pivots(rows[:1], cols[:1], 0.1)
        T1, _ = dt.download()
        r, q = int(rows[0]), int(chosen[0])
        e = np.zeros(R); e[r] = 1.0
        assert np.array_equal(T1[:, q], e)
        assert np.array_equal(_bits(T1[r]), _bits(T0[r] / T0[r, q]))
        other = (r + 1) % R
        assert np.array_equal(_bits(T1[other]), _bits(T0[other] - T0[other, q] * T1[r]))
        # same pivot again: factors are all zero -> nothing changes
        dt.forced_pivots(np.array([r], np.int32), np.array([q], np.int32), 0.1)
        T2, _ = dt.download()
        assert np.array_equal(_bits(T2), _bits(T1))
        # a few more pivots keep every previously pivoted column a unit vector unless its row is reused
        chosen2, _ = dt.forced_pivots(rows[1:], cols[1:], 0.1)
        T3, _ = dt.download()
        last_r, last_q = int(rows[-1]), int(chosen2[-1])
        e2 = np.zeros(R); e2[last_r] = 1.0
        assert np.array_equal(T3[:, last_q], e2)


def test_multi_run_batched_mixed_group(gpu, oracle):
    """lpx_multi_run (K9): a mixed group of primal and dual tableaux of different shapes advances in
    lockstep launches; every member must end bit-identical to its own sequential oracle run."""
    specs = [(40, 60, 2, 0), (64, 100, 3, 0), (8, 12, 1, 0), (20, 30, 2, 5), (100, 160, 5, 30), (30, 45, 21, 0), (40, 64, 3, 10)]
    tabs, refs, dual = [], [], []
    for (m, n, seed, n_ge) in specs:
        c, A, b = synth.dense_lp(m, n, seed=seed)
        T, basis = synth.primal_tableau_from(c, A, b)
        if n_ge:
            g = np.random.Generator(np.random.PCG64(seed + 99))
            for i in g.choice(m, size=n_ge, replace=False):
                T[i, :n] *= -1.0
                T[i, -1] = -0.02 * T[i, -1]
        Tr, br = T.copy(), basis.copy()
        if n_ge:
            st, tr, _ = oracle.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
        else:
            st, tr = oracle.primal_tableau(Tr, br)
        refs.append((st, tr, Tr, br))
        tabs.append(gpu.DeviceTableau.from_host(T, basis))
        dual.append(bool(n_ge))
    statuses, stats = gpu.multi_run(tabs, dual, dual_opts=gpu.default_opts(True, fdf_guard=10000, cleanup=1))
    for t, (st, tr, Tr, br), s, ss in zip(tabs, refs, statuses, stats):
        Tg, bg = t.download()
        assert s == st and ss["pivots"] == len(tr)
        assert t.trace().tolist() == tr.tolist()
        assert np.array_equal(_bits(Tg), _bits(Tr)) and bg.tolist() == br.tolist()
        t.close()
