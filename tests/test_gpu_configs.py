"""GPU parity AT THE SIZES OF BASELINE.json -- the shapes bench.py times are the shapes checked here.

* north-star shape 4096x8192 and the LP-level shape 4097x12289: forced Gauss-Jordan pivots (Models/PrimalSimplex.cs:245-257)
  and a real primal solve (ChooseEntering/ChooseLeaving/Pivot, :92-124), whole tableau bit-equal to the oracle;
* config 3 (revised, m=4096 n=8192): oracle-compared at m=1024 n=2048 (the oracle re-inverts an m x m basis per
  iteration, Models/RevisedPrimalSimplex.cs:128, 1.3 s each there), and at full size through size-independent
  properties of the maintained inverse (B^-1 B = I, z against a freshly refactored z);
* config 4 (0/1 IP n=512 m=256 + 512 bound rows): root + first DFS nodes of BranchAndBound.SolveNode
  (Models/Branch&Bound.cs:128-258), node log and node z bitwise, on the resident group kernel and on the streaming kernels;
* config 5 (100k-item knapsack): the first 4000 pops of BranchAndBoundKnapsack.Solve (Models/BranchAndBoundKnapsack.cs:118-328).

Everything goes through the C ABI (ctypes).  The oracle is the checker only (oracle/lpx_oracle.h: parity unpinned by
the reference, which ships no fixtures)."""
import hashlib
import os
import json
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("R,C,count", [(4096, 8192, 12), (4097, 12289, 8)])
def test_forced_pivots_bitwise_at_headline_shapes(gpu, oracle, R, C, count):
    """K4 on the shapes the roofline is quoted on (256 MiB: lpx_update_mb, cache policy default; 403 MB: lpx_update_mb_m,
    one row in three stored through the Infinity Cache, the entering column reduced by select's last workgroup): every
    element of the tableau after `count` pivots equals the oracle's (uint64 view), and so does every chosen pivot column."""
    T0 = synth.raw_tableau(R, C)
    rows, cols = synth.forced_pivot_list(R, C, count)
    Tr = T0.copy()
    chosen_ref = oracle.forced_pivots(Tr, rows, cols, 0.1)
    with gpu.DeviceTableau.from_host(T0) as dt:
        chosen, st = dt.forced_pivots(rows, cols, 0.1)
        Tg, _ = dt.download()
    assert chosen.tolist() == chosen_ref.tolist() and st["pivots"] == count
    assert np.array_equal(_bits(Tg), _bits(Tr))


@pytest.mark.parametrize("env", [{"LPX_UPDATE_POLICY": "1"}, {"LPX_UPDATE_MIXMOD": "2"}, {"LPX_UPDATE_MIXMOD": "5"}])
def test_every_streaming_update_form_is_bitwise(oracle, env):
    """The forms the launcher picks for still larger tableaux -- all non-temporal (lpx_update_mb_s) and the thinner store mixes
    (every 2nd / 5th row block keeps a row in the cache) -- forced onto the 403 MB tableau through the diagnostic knobs
    (read once per process, hence the child process): same bits as the oracle after 6 forced pivots."""
    R, C, count = 4097, 12289, 6
    T0 = synth.raw_tableau(R, C)
    rows, cols = synth.forced_pivot_list(R, C, count)
    Tr = T0.copy()
    chosen_ref = oracle.forced_pivots(Tr, rows, cols, 0.1)
    want = hashlib.sha256(np.ascontiguousarray(Tr).view(np.uint8)).hexdigest()
    code = textwrap.dedent(f"""
        import hashlib, numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        T0 = synth.raw_tableau({R}, {C}); rows, cols = synth.forced_pivot_list({R}, {C}, {count})
        with L.DeviceTableau.from_host(T0) as dt:
            chosen, st = dt.forced_pivots(rows, cols, 0.1)
            Tg, _ = dt.download()
        print(hashlib.sha256(np.ascontiguousarray(Tg).view(np.uint8)).hexdigest(), chosen.tolist() == {chosen_ref.tolist()!r}, st["pivots"])
    """)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=ROOT, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = r.stdout.strip().splitlines()[-1].split()
    assert got == [want, "True", str(count)], (env, got)


@pytest.mark.parametrize("cap", [150, 151])
def test_primal_solve_first_pivots_of_the_4096x8192_lp(gpu, oracle, cap):
    """The LP bench.py's headline value is measured on (m=4096, n=8192, tableau 4097x12289 = 403 MB, fused streaming
    kernel lpx_pivot_fused: update(k) out of place beside select(k+1)): first 150 / 151 pivots -- three hipGraph batches;
    an even and an odd count, so the tableau ends once in each of the two buffers -- trace / basis / whole tableau bit-equal."""
    c, A, b = synth.dense_lp(4096, 8192)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br, max_iter=cap)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.primal_run(max_iter=cap)
        assert st["launches"] > cap                     # streaming path: a launch per pivot, not one resident launch
        assert st["pivots"] == cap
        tr = dt.trace()
        Tg, bg = dt.download()
    assert status == st_ref == 3 and len(tr_ref) == cap
    assert tr.tolist() == tr_ref.tolist() and bg.tolist() == br.tolist()
    assert np.array_equal(_bits(Tg), _bits(Tr))


_FULL_SOLVE = textwrap.dedent("""
    import hashlib, json, numpy as np
    import linear_programming_solver_lpr381_amd as L
    from linear_programming_solver_lpr381_amd import synth
    c, A, b = synth.dense_lp(4096, 8192)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).view(np.uint8)).hexdigest()
    with L.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.primal_run(max_iter=200000)
        Tg, bg = dt.download()
        tr = dt.trace()
    print(json.dumps([int(status), int(st["pivots"]), int(st["launches"]), h(tr), h(bg), h(Tg), float(Tg[-1, -1])]))
""")


def test_headline_lp_solved_to_optimality_same_bits_on_both_streaming_paths():
    """The headline LP (m=4096 n=8192, tableau 403 MB) solved to OPTIMAL -- about 80 000 pivots, which the CPU oracle would
    need half an hour for -- once on the fused launch (lpx_pivot_fused, out of place) and once on the two-launch in-place
    kernels that the 150-pivot test above pins to the oracle: status, pivot count, the recorded pivot trace, basis and every
    bit of the final tableau agree."""
    runs = {}
    for tag, env in (("fused", {}), ("two-launch", {"LPX_FUSED_PIVOT": "0"})):
        r = subprocess.run([sys.executable, "-c", _FULL_SOLVE], env=dict(os.environ, PYTHONPATH=ROOT, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        runs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    f, t = runs["fused"], runs["two-launch"]
    assert f[0] == t[0] == 0 and f[1] == t[1] > 50000          # OPTIMAL, same number of pivots
    assert f[2] < 1.1 * f[1] and t[2] > 1.9 * t[1]             # one launch per pivot against two
    assert f[3:] == t[3:], (f, t)


def _small_lps():
    """(name, T, basis): LPs solved to the end -- optimal, unbounded, degenerate ties, and a second run on the same handle."""
    out = []
    # row counts around the fused select's batch (6 x 256 rows) and wave (64) boundaries, columns below one workgroup's slice
    for m, n, seed in [(48, 80, 3), (200, 320, 5), (130, 64, 7), (255, 24, 11), (256, 24, 12), (1535, 20, 13), (1536, 20, 14), (1537, 20, 15)]:
        c, A, b = synth.dense_lp(m, n, seed=seed)
        T, basis = synth.primal_tableau_from(c, A, b)
        out.append((f"dense{m}x{n}", T, basis))
    c = np.array([1.0, 1.0]); A = np.array([[1.0, -1.0], [-1.0, 0.5]]); b = np.array([1.0, 2.0])
    T, basis = synth.primal_tableau_from(c, A, b)
    out.append(("unbounded", T, basis))
    c = np.array([3.0, 2.0, 1.0]); A = np.array([[1.0, 1.0, 0.0], [1.0, 0.0, 1.0], [1.0, 1.0, 1.0], [2.0, 1.0, 0.0]])
    b = np.array([4.0, 4.0, 4.0, 8.0])
    T, basis = synth.primal_tableau_from(c, A, b)
    out.append(("ties", T, basis))
    T, basis = synth.primal_tableau_from(np.array([2.0]), np.array([[4.0]]), np.array([3.0]))      # one row, one column: 2 x 3
    out.append(("one-by-one", T, basis))
    T, basis = synth.primal_tableau_from(np.array([1.0, 3.0]), np.array([[1.0, 2.0], [3.0, 1.0]]), np.array([4.0, 6.0]))
    out.append(("two-by-two", T, basis))
    return out


@pytest.mark.parametrize("env", [{}, {"LPX_FUSED_PIVOT": "0"}, {"LPX_UPDATE_POLICY": "1"}, {"LPX_UPDATE_POLICY": "0"}, {"LPX_UPDATE_POLICY": "0", "LPX_FUSED_PIVOT": "0"}],
                         ids=["fused", "two-launch", "fused-all-nt", "fused-cached", "two-launch-cached"])
def test_streaming_primal_loop_to_the_end_on_small_lps(oracle, env):
    """The streaming primal loop run to its END -- optimal, unbounded, the iteration cap hit exactly at the optimum's pivot
    count -- which the 403 MB LP is too long for: every cache-policy form is forced onto small tableaux (LPX_UPDATE_POLICY,
    resident kernels off; knobs read once per process, hence the child).  Fused (one launch per pivot: streaming mix, all
    non-temporal, default policy) and two-launch in-place paths (LPX_FUSED_PIVOT=0: what runs under a per-pivot callback):
    status, pivot trace, basis and every bit of the tableau as the oracle's."""
    want = []
    for name, T, basis in _small_lps():
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = oracle.primal_tableau(Tr, br, max_iter=10000)
        caps = [10000] + ([len(tr_ref), len(tr_ref) - 1] if len(tr_ref) > 1 else [])
        for cap in caps:
            Tc, bc = T.copy(), basis.copy()
            s_c, t_c = oracle.primal_tableau(Tc, bc, max_iter=cap)
            want.append([name, cap, int(s_c), len(t_c), hashlib.sha256(np.ascontiguousarray(t_c, dtype=np.int32).view(np.uint8)).hexdigest(),
                         hashlib.sha256(np.ascontiguousarray(bc, dtype=np.int32).view(np.uint8)).hexdigest(),
                         hashlib.sha256(np.ascontiguousarray(Tc).view(np.uint8)).hexdigest()])
    code = textwrap.dedent("""
        import hashlib, json, sys, numpy as np
        sys.path.insert(0, %r)
        import linear_programming_solver_lpr381_amd as L
        from test_gpu_configs import _small_lps
        h = lambda a, dt: hashlib.sha256(np.ascontiguousarray(a, dtype=dt).view(np.uint8)).hexdigest()
        out = []
        for name, T, basis in _small_lps():
            caps = json.loads(sys.argv[1])[name]
            with L.DeviceTableau.from_host(T, basis) as dt:
                dt.snapshot()
                for cap in caps:
                    dt.restore()
                    status, st = dt.primal_run(max_iter=cap, resident=-1)
                    assert st["launches"] >= st["pivots"], st
                    Tg, bg = dt.download()
                    out.append([name, cap, int(status), int(st["pivots"]), h(dt.trace(), np.int32), h(bg, np.int32), h(Tg, np.float64)])
        print(json.dumps(out))
    """ % os.path.join(ROOT, "tests"))
    caps = {}
    for w in want:
        caps.setdefault(w[0], []).append(w[1])
    e = dict(os.environ, PYTHONPATH=ROOT, LPX_RESIDENT="0", **{"LPX_UPDATE_POLICY": "2", **env})
    r = subprocess.run([sys.executable, "-c", code, json.dumps(caps)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got == want, [(g, w) for g, w in zip(got, want) if g != w]


def test_config3_revised_vs_oracle_m1024(gpu, oracle):
    """Revised loop (Models/RevisedPrimalSimplex.cs:66-142) against the reference-faithful oracle (full Invert per
    iteration) at m=1024 n=2048: pivots, Bidx, Nidx list order equal; z within 1e-9 relative."""
    m, n, iters = 1024, 2048, 10
    c, A, b = synth.dense_lp(m, n)
    ref = oracle.revised_solve(oracle.Problem(oracle.MAX, c, A, np.zeros(m, np.int32), b), max_iter=iters)
    with gpu.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run(max_iter=iters)
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
    assert status == ref.status == 3 and st["pivots"] == iters
    assert tr.tolist() == ref.trace.tolist()
    assert Bidx.tolist() == ref.Bidx.tolist() and Nidx.tolist() == ref.Nidx.tolist()
    assert abs(z - ref.z_internal) <= 1e-9 * abs(ref.z_internal)
    assert np.allclose(xB, ref.xB, rtol=1e-9, atol=1e-9)


def test_config3_full_size_inverse_stays_an_inverse(gpu):
    """Config 3 at full size (m=4096 n=8192): after the 300 product-form iterations bench.py times, the maintained
    B^-1 is still the inverse of the basis matrix (max |B^-1 B - I| <= 1e-8), x_B = B^-1 b, and z agrees with the z
    of a fresh refactorisation (K7', the reference's own Invert) to 1e-9 relative."""
    m, n, iters = 4096, 8192, 300
    c, A, b = synth.dense_lp(m, n)
    with gpu.DeviceRevised(A, -c, b) as rv:
        status, st = rv.run(max_iter=iters, batch=50)
        assert status == 3 and st["pivots"] == iters
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
        Binv = rv.binv()
        rv.refactor()
        Bidx2, Nidx2, xB2, z2 = rv.result()
        Binv2 = rv.binv()
    assert len(set(Bidx.tolist())) == m and sorted(Bidx.tolist() + Nidx.tolist()) == list(range(n + m))
    assert (tr[:, 0] >= 0).all() and (tr[:, 0] < m).all() and (tr[:, 1] >= 0).all() and (tr[:, 1] < n + m).all()
    Bm = np.empty((m, m))
    for k, j in enumerate(Bidx):
        Bm[:, k] = A[:, j] if j < n else np.eye(1, m, j - n).ravel()
    E = Binv @ Bm
    E[np.arange(m), np.arange(m)] -= 1.0
    assert np.abs(E).max() <= 1e-8, np.abs(E).max()
    assert np.allclose(Binv @ b, xB, rtol=1e-9, atol=1e-9)
    assert Bidx2.tolist() == Bidx.tolist() and Nidx2.tolist() == Nidx.tolist()
    assert abs(z - z2) <= 1e-9 * abs(z2), (z, z2)
    assert np.allclose(xB, xB2, rtol=1e-8, atol=1e-8)
    E2 = Binv2 @ Bm
    E2[np.arange(m), np.arange(m)] -= 1.0
    assert np.abs(E2).max() <= 1e-9
    cfull = np.concatenate([-c, np.zeros(m)])
    assert abs(cfull[Bidx] @ xB2 - z2) <= 1e-9 * abs(z2)


_CONFIG4 = textwrap.dedent('''
    import numpy as np
    import linear_programming_solver_lpr381_amd as L
    from linear_programming_solver_lpr381_amd import synth
    from oracle import oracle as O
    L._lib.check(L._lib.lib().lpx_init(0))
    c, A, rel, b = synth.binary_ip(512, 256)
    p = L.LPProblem.from_arrays(0, c, A, rel, b)
    po = O.Problem(O.MAX, c, A, rel.astype(np.int32), b)
    for mode in (0, 1):
        ref = O.bnb_solve(po, mode, max_nodes=5)
        r = L.BranchAndBound(bnb_mode=mode, max_nodes=5).Solve(p)
        assert r.NodeLog.tolist() == ref.log.tolist(), (mode, r.NodeLog.tolist(), ref.log.tolist())
        assert np.array_equal(r.NodeZ.view(np.uint64), ref.log_z.view(np.uint64)), mode
        assert r.LpSolves == ref.lp_solves == 6 and r.Nodes == ref.nodes_visited == 5
        assert r.Stats["pivots"] == ref.total_pivots, (mode, r.Stats["pivots"], ref.total_pivots)
        assert r.Tableau.shape == (769, 1281)
    print("OK")
''')


@pytest.mark.parametrize("env", [{}, {"LPX_RESIDENT_GROUP": "0"}, {"LPX_RESIDENT_GROUP": "0", "LPX_RESIDENT": "0"}],
                         ids=["resident", "group-off", "streaming-only"])
def test_config4_root_and_first_dfs_nodes_vs_oracle(oracle, env):
    """Config 4 at full size: the root LP (twice, Models/Branch&Bound.cs:57,:95) and the first nodes of the DFS --
    node log (depth, outcome, branching variable), node z bitwise, LP and pivot counts -- in faithful and repaired
    mode, with the node LPs on the resident group kernel (default), on the batched/streaming kernels
    (LPX_RESIDENT_GROUP=0) and with the root on the streaming kernels too (LPX_RESIDENT=0)."""
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    r = subprocess.run([sys.executable, "-c", _CONFIG4], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_config5_first_4000_pops_vs_oracle(gpu, oracle):
    """Config 5 at full size (n=100000): popped nodes, relaxations, expansions, incumbent of the first 4000 pops
    equal the oracle's best-first loop (Models/BranchAndBoundKnapsack.cs:118-328, heap :494-547)."""
    p, w, cap = synth.knapsack(100_000)
    ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=4000)
    kp = gpu.LPProblem(gpu.Sense.Max, p.tolist(), [gpu.Constraint(w.tolist(), gpu.Rel.LE, cap)])
    r = gpu.BranchAndBoundKnapsack(max_nodes=4000).Solve(kp)
    assert r.Nodes == ref.nodes_popped == 4000
    assert r.Aux[0] == ref.relaxations and r.Aux[2] == ref.nodes_expanded and r.Aux[3] == ref.max_heap
    assert (r.OptimalValue == ref.best_z) or (np.isinf(ref.best_z) and np.isinf(r.OptimalValue))
    if ref.best_x.size and np.isfinite(ref.best_z):
        assert r.Extra.astype(int).tolist() == ref.best_x.tolist()


def test_config5_root_bound_and_order(gpu, oracle):
    """Config 5: the ratio order (stable sort, Models/BranchAndBoundKnapsack.cs:75-79) and the root relaxation
    (:102-113) of the 100k-item instance equal the oracle's; a few hundred deep nodes too."""
    p, w, cap = synth.knapsack(100_000)
    dk = gpu.DeviceKnapsack(p, w, cap)
    order = oracle.knapsack_order(p, w)
    assert hashlib.sha256(dk.order().tobytes()).hexdigest() == hashlib.sha256(order.tobytes()).hexdigest()
    g = np.random.default_rng(12)
    nodes = [{}]
    for depth in (1, 7, 100, 250, 257, 900):
        idx = g.choice(100_000, size=depth, replace=False)
        nodes.append({int(i): int(g.integers(0, 2)) for i in idx})
    P, W, F, X = dk.relax_batch(nodes)
    for j, nd in enumerate(nodes):
        a = -np.ones(100_000, np.int32)
        for i, v in nd.items():
            a[i] = v
        rp, rw, rf, rx = oracle.knapsack_relax(p, w, cap, order, a, want_vector=True)
        assert (P[j], W[j], F[j]) == (rp, rw, rf), j
        if rf >= 0:
            assert X[j] == rx[order[rf]]
    dk.close()


def test_dual_path_on_a_tableau_larger_than_the_infinity_cache(gpu, oracle):
    """lpx_update_s (the streaming variant of the dual / forced path's update kernel: 3 rows per wave, non-temporal loads and
    stores) only runs above 292 MiB: a 4501 x 10001 dual tableau (361 MB), first pivots of ForceDualFeasibility + dual loop +
    clean-up, bit-equal to the oracle."""
    m, n = 4500, 5500
    c, A, b = synth.dense_lp(m, n, seed=11)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    g = np.random.Generator(np.random.PCG64(11))
    for i in g.choice(m, size=12, replace=False):
        T[i, :n] *= -1.0
        T[i, -1] = -0.02 * T[i, -1]
    assert T.nbytes > (292 << 20)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref, nf = oracle.dual_tableau(Tr, br, fdf_guard=4, cleanup=1, max_iter=6)
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.dual_run(fdf_guard=4, cleanup=1, max_iter=6)
        tr = dt.trace()
        Tg, bg = dt.download()
    assert status == st_ref and tr.tolist() == tr_ref.tolist() and len(tr) >= 6 and st["fdf_pivots"] == nf
    assert bg.tolist() == br.tolist() and np.array_equal(_bits(Tg), _bits(Tr))
