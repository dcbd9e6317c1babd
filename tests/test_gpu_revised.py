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
