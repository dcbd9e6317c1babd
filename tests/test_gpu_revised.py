"""GPU parity: revised primal simplex (rank-1 updated B^-1) against the oracle's faithful restatement
(full re-inversion per iteration).  The two round differently by construction, so the bar is the one
north_star states: identical pivot sequence and basis (integer work, bit-exact) on instances without
near-ties, objective within 1e-9 relative."""
import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu
REL = 1e-9


def test_kat2_revised(gpu):
    A = np.array([[1.0, 0], [0, 2], [3, 2]]); b = np.array([4.0, 12, 18]); C_ = np.array([3.0, 5.0])
    with gpu.DeviceRevised(A, -C_, b) as rv:
        status, st = rv.run()
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
    assert status == 0 and tr.tolist() == [[1, 1], [2, 0]]
    assert Bidx.tolist() == [2, 1, 0]
    assert xB.tolist() == [2.0, 6.0, 2.0] and z == -36.0
    assert Nidx.tolist() == [3, 4]        # RemoveAt + Add order: c2 then c3 (Models/RevisedPrimalSimplex.cs:123-124)


@pytest.mark.parametrize("m,n,seed", [(8, 12, 1), (20, 30, 7), (40, 64, 3), (64, 100, 4), (96, 160, 9), (128, 256, 4)])
def test_random_lp_same_pivots_as_oracle(gpu, oracle, m, n, seed):
    c, A, b = synth.dense_lp(m, n, seed=seed)
    p = oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b)
    ref = oracle.revised_solve(p)
    with gpu.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run()
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
        Binv = rv.binv()
    assert status == ref.status == 0
    assert tr.tolist() == ref.trace.tolist()
    assert Bidx.tolist() == ref.Bidx.tolist()
    assert Nidx.tolist() == ref.Nidx.tolist()
    assert abs(z - ref.z_internal) <= REL * abs(ref.z_internal)
    assert np.allclose(xB, ref.xB, rtol=1e-9, atol=1e-9)
    # the maintained inverse really is the inverse of the final basis
    full = np.hstack([A, np.eye(m)])
    assert np.allclose(Binv @ full[:, Bidx], np.eye(m), atol=1e-8)
    x = np.zeros(n); x[Bidx[Bidx < n]] = xB[Bidx < n]
    assert abs(c @ x - ref.z_original) <= REL * abs(ref.z_original)


def test_revised_iteration_limit_and_unbounded(gpu, oracle):
    c, A, b = synth.dense_lp(20, 30, seed=7)
    p = oracle.Problem(oracle.MAX, c, A, np.zeros(20, np.int32), b)
    ref = oracle.revised_solve(p, max_iter=3)
    assert ref.status == oracle.ITER_LIMIT
    with gpu.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run(max_iter=3, batch=2)
        assert status == 3 and rv.trace().tolist() == ref.trace.tolist()
    # unbounded: max x1 s.t. -x1 + x2 <= 1
    with gpu.DeviceRevised(np.array([[-1.0, 1.0]]), -np.array([1.0, 0.0]), np.array([1.0])) as rv:
        status, _ = rv.run()
    assert status == 1


def test_revised_one_shot(gpu, oracle):
    import ctypes as C
    from linear_programming_solver_lpr381_amd._lib import dp, ip, lib, Stats, NULL_CB, check
    m, n = 30, 45
    c, A, b = synth.dense_lp(m, n, seed=21)
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b))
    cm = np.ascontiguousarray(-c)
    Bidx = np.zeros(m, np.int32); Nidx = np.zeros(n, np.int32); xB = np.zeros(m); z = C.c_double(); st = Stats()
    rc = check(lib().lpx_revised_solve(np.ascontiguousarray(A).ctypes.data_as(dp), m, n, cm.ctypes.data_as(dp),
                                       b.ctypes.data_as(dp), Bidx.ctypes.data_as(ip), Nidx.ctypes.data_as(ip),
                                       xB.ctypes.data_as(dp), C.byref(z), 1e-9, 10000, NULL_CB, None, C.byref(st)))
    assert rc == 0 and Bidx.tolist() == ref.Bidx.tolist() and st.pivots == len(ref.trace)
    assert abs(z.value - ref.z_internal) <= REL * abs(ref.z_internal)


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (12, 3), (100, 4), (257, 5), (600, 6)])
def test_invert_bitwise_vs_oracle(gpu, oracle, n, seed):
    """K7' / SURVEY a19: Invert (Gauss-Jordan, partial pivoting, first max on ties) bit for bit."""
    g = np.random.default_rng(seed)
    M = g.uniform(-1, 1, size=(n, n))
    if n > 2:
        M[0, :] = M[1, :] * 0 + M[0, :]          # keep generic
        M[2, 0] = -M[1, 0]                       # a tie in |a| on the first column: the first row must win
    rc, ref = oracle.invert(M)
    assert rc == 0
    inv = gpu.invert(M)
    assert np.array_equal(inv.view(np.uint64), ref.view(np.uint64))
    assert np.allclose(inv @ M, np.eye(n), atol=1e-8)


def test_invert_singular(gpu, oracle):
    M = np.ones((5, 5))
    assert oracle.invert(M)[0] == oracle.E_SINGULAR
    with pytest.raises(gpu.LpxError) as e:
        gpu.invert(M)
    assert e.value.code == gpu._lib.E_SINGULAR and "Singular basis encountered." in str(e.value)


@pytest.mark.parametrize("every", [1, 3, 7])
def test_revised_with_periodic_refactorisation(gpu, oracle, every):
    """With refactor_every = 1 the engine recomputes B^-1 from the basis after every iteration, as the
    reference does (RevisedPrimalSimplex.cs:128); any interval must give the oracle's pivots."""
    m, n, seed = 40, 64, 3
    c, A, b = synth.dense_lp(m, n, seed=seed)
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b))
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.set_refactor(every)
        status, st = rv.run()
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
        Binv = rv.binv()
    assert status == 0 and tr.tolist() == ref.trace.tolist() and st["pivots"] == len(ref.trace)
    assert Bidx.tolist() == ref.Bidx.tolist() and Nidx.tolist() == ref.Nidx.tolist()
    assert abs(z - ref.z_internal) <= REL * abs(ref.z_internal)
    full = np.hstack([A, np.eye(m)])
    assert np.allclose(Binv @ full[:, Bidx], np.eye(m), atol=1e-9)


def test_refactor_restores_exact_inverse(gpu, oracle):
    m, n = 30, 45
    c, A, b = synth.dense_lp(m, n, seed=21)
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.run()
        Bidx, _, xB0, z0 = rv.result()
        rv.refactor()
        _, _, xB1, z1 = rv.result()
        Binv = rv.binv()
    full = np.hstack([A, np.eye(m)])
    rc, ref = oracle.invert(np.ascontiguousarray(full[:, Bidx]))
    assert rc == 0 and np.array_equal(Binv.view(np.uint64), ref.view(np.uint64))     # same Invert, same bits
    assert np.allclose(xB1, xB0, rtol=1e-10, atol=1e-10) and abs(z1 - z0) <= 1e-10 * abs(z0)


def test_refactor_segments_fire_each_pivot_callback_once(gpu, oracle):
    """A run split into segments by the periodic refactorisation keeps ONE iteration count: the trace is not
    restarted and every pivot's callback fires exactly once, in order (LoopRun::begin takes `init.iter` as the
    number of callbacks already fired).  Round-1 record gpurun_out/t41.log was this path with the count reset
    to 0 per segment."""
    m, n, seed = 40, 64, 3
    c, A, b = synth.dense_lp(m, n, seed=seed)
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b))
    ev = []
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.set_refactor(3)
        status, st = rv.run(cb=lambda it, r, q: ev.append((it, r, q)), batch=2)
        tr = rv.trace()
    assert status == 0 and st["pivots"] == len(ref.trace) and tr.tolist() == ref.trace.tolist()
    assert [e[0] for e in ev] == list(range(1, len(tr) + 1))
    assert [[e[1], e[2]] for e in ev] == tr.tolist()


@pytest.mark.parametrize("m,n,seed", [(1, 3, 1), (2, 5, 2), (12, 20, 3), (40, 64, 3), (100, 150, 4), (257, 300, 5)])
def test_fast_refactor_newton_schulz_on_matrix_cores(gpu, oracle, m, n, seed):
    """lpx_revised_set_refactor_mode(1): B^-1 refreshed by Newton-Schulz steps whose two m x m x m contractions run on
    v_mfma_f64_16x16x4_f64 (csrc/lpx_mfma.hip), any m (ragged tiles).  Bar: |B^-1 B - I| <= 1e-12 here, the same inverse
    as the exact Gauss-Jordan to 1e-10, x_B / z to 1e-9."""
    c, A, b = synth.dense_lp(m, n, seed=seed)
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.run()
        Bidx, _, xB0, z0 = rv.result()
        rv.set_refactor_mode(1)
        rv.refactor()
        _, _, xB1, z1 = rv.result()
        Binv1 = rv.binv()
        st = rv.refactor_stats()
        rv.set_refactor_mode(0)
        rv.refactor()
        Binv0 = rv.binv()
        _, _, xB2, z2 = rv.result()
    assert st["refactors"] == 1 and st["fast_steps"] >= 1 and st["fast_fallbacks"] == 0
    full = np.hstack([A, np.eye(m)])
    E = Binv1 @ full[:, Bidx] - np.eye(m)
    assert np.abs(E).max() <= 1e-12, np.abs(E).max()
    assert np.allclose(Binv1, Binv0, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(Binv0).max()))
    assert np.allclose(xB1, xB2, rtol=1e-9, atol=1e-9) and abs(z1 - z2) <= 1e-9 * max(1.0, abs(z2))
    assert np.allclose(xB1, xB0, rtol=1e-9, atol=1e-9) and abs(z1 - z0) <= 1e-9 * max(1.0, abs(z0))


@pytest.mark.parametrize("every", [1, 4])
def test_fast_refactor_keeps_the_oracle_pivots(gpu, oracle, every):
    """Periodic refactorisation in fast mode: the pivot sequence, Bidx and Nidx order stay those of the oracle."""
    m, n, seed = 40, 64, 3
    c, A, b = synth.dense_lp(m, n, seed=seed)
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b))
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.set_refactor_mode(1)
        rv.set_refactor(every)
        status, st = rv.run()
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
        stats = rv.refactor_stats()
    assert status == 0 and tr.tolist() == ref.trace.tolist()
    assert Bidx.tolist() == ref.Bidx.tolist() and Nidx.tolist() == ref.Nidx.tolist()
    assert abs(z - ref.z_internal) <= REL * abs(ref.z_internal)
    assert stats["refactors"] >= len(tr) // every - 1 and stats["fast_fallbacks"] == 0


def test_drift_policy_checks_the_residual_and_refactors_only_when_asked(gpu, oracle):
    """Default drift control (include/lpx.h): every `check_every` iterations the residual rho of the maintained inverse is
    evaluated on the device; a refactorisation happens only above `tol`.  m = 1024: with the default tolerance nothing is
    refactored and rho stays tiny; with tol = 0 every check refactors (fast mode) and the first 10 pivots are still the
    oracle's (it re-inverts every iteration, Models/RevisedPrimalSimplex.cs:128)."""
    m, n = 1024, 2048
    c, A, b = synth.dense_lp(m, n)
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.set_drift_policy(100, 1e-9)
        status, st = rv.run(max_iter=450)
        s0 = rv.refactor_stats()
        rho, absr = rv.residual()
        Bidx, _, xB, z = rv.result()
        tr0 = rv.trace()
    assert status == 3 and st["pivots"] == 450
    assert s0["refactors"] == 0 and 0.0 <= s0["last_residual"] <= 1e-10 and rho <= 1e-10
    full_x = np.zeros(n + m); full_x[Bidx] = xB
    assert np.abs(np.hstack([A, np.eye(m)]) @ full_x - b).max() <= 1e-9 * (1 + np.abs(b).max())
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b), max_iter=10)
    with gpu.DeviceRevised(A, -c, b) as rv:
        rv.set_refactor_mode(1)
        rv.set_drift_policy(3, 0.0)                      # every check refactors
        status, st = rv.run(max_iter=10)
        s1 = rv.refactor_stats()
        tr = rv.trace()
        Bidx1, Nidx1, _, z1 = rv.result()
    assert s1["refactors"] == 3 and s1["fast_fallbacks"] == 0
    assert tr.tolist() == ref.trace.tolist() == tr0[:10].tolist()
    assert Bidx1.tolist() == ref.Bidx.tolist() and Nidx1.tolist() == ref.Nidx.tolist()
    assert abs(z1 - ref.z_internal) <= REL * abs(ref.z_internal)


def test_five_launch_path_still_matches_the_oracle(oracle):
    """LPX_REVISED_FUSED=0: the round-1 iteration (separate pricing / pick / FTRAN / select / eager W update), which is also
    the path of bases with more than 8192 rows -- same pivots, Bidx, Nidx as the oracle and as the fused path."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np, json
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        out = []
        for (m, n, seed) in [(20, 30, 7), (96, 160, 9), (257, 300, 5)]:
            c, A, b = synth.dense_lp(m, n, seed=seed)
            with L.DeviceRevised(A, -c, b) as rv:
                status, st = rv.run()
                Bidx, Nidx, xB, z = rv.result()
                out.append([status, rv.trace().tolist(), Bidx.tolist(), Nidx.tolist(), z])
        print(json.dumps(out))
    ''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for fused in ("1", "0"):
        env = dict(os.environ, LPX_REVISED_FUSED=fused, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        import json
        res[fused] = json.loads(r.stdout.strip().splitlines()[-1])
    for k, (m, n, seed) in enumerate([(20, 30, 7), (96, 160, 9), (257, 300, 5)]):
        c, A, b = synth.dense_lp(m, n, seed=seed)
        ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b))
        for fused in ("1", "0"):
            status, tr, Bidx, Nidx, z = res[fused][k]
            assert status == ref.status == 0 and tr == ref.trace.tolist(), (fused, m)
            assert Bidx == ref.Bidx.tolist() and Nidx == ref.Nidx.tolist()
            assert abs(z - ref.z_internal) <= REL * abs(ref.z_internal)
