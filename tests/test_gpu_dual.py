"""GPU parity: the HIP dual path (ForceDualFeasibility + dual loop + repaired clean-up) against the
CPU oracle, through the C ABI.  Bit-exact tableaux, traces and bases."""
import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _dual_tableau(m, n, seed, n_ge, scale_b=1.0):
    """All-<= tableau with `n_ge` rows turned into `-row <= -b'` (a repaired >= row)."""
    c, A, b = synth.dense_lp(m, n, seed=seed)
    T, basis = synth.primal_tableau_from(c, A, b)
    g = np.random.Generator(np.random.PCG64(seed + 99))
    rows = g.choice(m, size=n_ge, replace=False)
    for i in rows:
        T[i, :n] *= -1.0
        T[i, -1] = -scale_b * 0.02 * T[i, -1]      # small positive demand -> feasible mixes
    return T, basis


@pytest.mark.parametrize("m,n,seed,n_ge", [(6, 8, 1, 2), (20, 30, 2, 5), (40, 64, 3, 10), (64, 100, 4, 7),
                                            (100, 160, 5, 30)])
@pytest.mark.parametrize("guard,cleanup", [(100, 0), (10000, 1), (3, 0), (3, 1)])
def test_dual_trace_and_tableau_bitwise(gpu, oracle, m, n, seed, n_ge, guard, cleanup):
    T, basis = _dual_tableau(m, n, seed, n_ge)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref, nfdf_ref = oracle.dual_tableau(Tr, br, fdf_guard=guard, cleanup=cleanup)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.dual_run(fdf_guard=guard, cleanup=cleanup)
        Tg, bg = dt.download()
        tr = dt.trace()
    assert status == st_ref
    assert tr.tolist() == tr_ref.tolist()
    assert st["fdf_pivots"] == nfdf_ref
    assert bg.tolist() == br.tolist()
    assert np.array_equal(_bits(Tg), _bits(Tr))


def test_dual_infeasible(gpu, oracle):
    # x1 <= 1 and x1 >= 2 (as -x1 <= -2): dual loop finds no entering column
    T = np.array([[1.0, 1, 0, 1], [-1.0, 0, 1, -2], [-1.0, 0, 0, 0]])
    basis = np.array([1, 2], dtype=np.int32)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref, _ = oracle.dual_tableau(Tr, br)
    assert st_ref == 2
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, _ = dt.dual_run()
        Tg, bg = dt.download()
        assert dt.trace().tolist() == tr_ref.tolist()
    assert status == 2
    assert np.array_equal(_bits(Tg), _bits(Tr))


def test_dual_one_shot(gpu, oracle):
    T, basis = _dual_tableau(30, 40, 8, 6)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref, _ = oracle.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
    status, st = gpu.dual_tableau(T, basis, fdf_guard=10000, cleanup=1)
    assert status == st_ref and st["pivots"] == len(tr_ref)
    assert np.array_equal(_bits(T), _bits(Tr)) and basis.tolist() == br.tolist()
