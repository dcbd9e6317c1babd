"""GPU parity of the on-chip resident primal loop (csrc/lpx_resident.hip: tableau in LDS, one persistent
workgroup per CU, tagged-granule exchange) against the CPU oracle AND against the streaming kernels:
bit-exact tableaux, traces, bases and statuses."""
import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _run(gpu, T, basis, **kw):
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        status, st = dt.primal_run(**kw)
        Tg, bg = dt.download()
        return status, st, Tg, bg, dt.trace()


# m below, at and above the CU count (1 row per workgroup / 2 rows per workgroup / ragged last workgroup),
# odd widths (padding lanes), one-row and one-column problems
@pytest.mark.parametrize("m,n,seed", [(1, 1, 1), (1, 7, 2), (3, 2, 3), (17, 9, 4), (64, 100, 5), (255, 300, 6),
                                      (256, 256, 7), (257, 401, 8), (300, 700, 9), (513, 1100, 10)])
def test_resident_matches_oracle_and_streaming_bitwise(gpu, oracle, m, n, seed):
    c, A, b = synth.dense_lp(m, n, seed=seed)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    s1, st1, T1, b1, tr1 = _run(gpu, T, basis, resident=1)
    s0, st0, T0, b0, tr0 = _run(gpu, T, basis, resident=-1)
    assert s1 == st_ref == s0
    assert tr1.tolist() == tr_ref.tolist() == tr0.tolist()
    assert b1.tolist() == br.tolist()
    assert np.array_equal(_bits(T1), _bits(Tr)) and np.array_equal(_bits(T0), _bits(Tr))
    assert st1["pivots"] == len(tr_ref) and st1["launches"] == 1


def test_resident_config2_full_solve_bitwise(gpu, oracle):
    c, A, b = synth.dense_lp(1024, 2048)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    s1, st1, T1, b1, tr1 = _run(gpu, T, basis, resident=1)
    assert s1 == st_ref == 0 and tr1.tolist() == tr_ref.tolist() and b1.tolist() == br.tolist()
    assert np.array_equal(_bits(T1), _bits(Tr))


def test_resident_callbacks_chunks_and_restarts(gpu, oracle):
    c, A, b = synth.dense_lp(96, 150, seed=21)
    T, basis = synth.primal_tableau_from(c, A, b)
    Tr, br = T.copy(), basis.copy()
    st_ref, tr_ref = oracle.primal_tableau(Tr, br)
    events = []
    with gpu.DeviceTableau.from_host(T, basis) as dt:
        dt.snapshot()
        status, st = dt.primal_run(resident=1, batch=32, cb=lambda it, r, q: events.append((it, r, q)))
        assert status == 0 and st["launches"] == -(-(len(tr_ref) + 1) // 32)
        assert [(e[1], e[2]) for e in events] == [tuple(x) for x in tr_ref.tolist()]
        assert [e[0] for e in events] == list(range(1, len(tr_ref) + 1))
        T1, _ = dt.download()
        assert np.array_equal(_bits(T1), _bits(Tr))
        for _ in range(3):                       # generations keep counting across runs of one handle
            dt.restore()
            status, st = dt.primal_run(resident=1)
            T2, b2 = dt.download()
            assert status == 0 and np.array_equal(_bits(T2), _bits(Tr)) and b2.tolist() == br.tolist()
        dt.restore()
        status, st = dt.primal_run(resident=-1)  # and the streaming path still starts cleanly afterwards
        T3, _ = dt.download()
        assert np.array_equal(_bits(T3), _bits(Tr))


def test_resident_terminal_statuses(gpu, oracle):
    # unbounded: max x1, -x1 + x2 <= 1
    T, basis = synth.primal_tableau_from(np.array([1.0, 0.0]), np.array([[-1.0, 1.0]]), np.array([1.0]))
    Tr, br = T.copy(), basis.copy()
    st_ref, _ = oracle.primal_tableau(Tr, br)
    s, _, Tg, _, _ = _run(gpu, T, basis, resident=1)
    assert s == st_ref == 1 and np.array_equal(_bits(Tg), _bits(Tr))
    # iteration limit: the k pivots are done, then the limit fires before ChooseEntering (:95-98)
    c, A, b = synth.dense_lp(40, 60, seed=2)
    T, basis = synth.primal_tableau_from(c, A, b)
    for k in (0, 1, 5):
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = oracle.primal_tableau(Tr, br, max_iter=k)
        s, st, Tg, bg, tr = _run(gpu, T, basis, resident=1, max_iter=k)
        assert s == st_ref == 3 and tr.tolist() == tr_ref.tolist() and np.array_equal(_bits(Tg), _bits(Tr))
    # already optimal: no pivot, one launch
    T, basis = synth.primal_tableau_from(np.array([-1.0, -2.0]), np.array([[1.0, 1.0]]), np.array([3.0]))
    s, st, Tg, _, tr = _run(gpu, T, basis, resident=1)
    assert s == 0 and len(tr) == 0 and np.array_equal(_bits(Tg), _bits(T))


def test_resident_degenerate_ties_follow_the_hysteresis_scan(gpu, oracle):
    g = np.random.default_rng(5)
    for trial in range(6):
        m, n = 40, 30
        A = g.integers(0, 4, size=(m, n)).astype(float)
        b = g.integers(0, 3, size=m).astype(float)            # many zero and equal ratios
        c = g.integers(1, 5, size=n).astype(float)
        T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = oracle.primal_tableau(Tr, br, max_iter=500)
        s, _, Tg, bg, tr = _run(gpu, T, basis, resident=1, max_iter=500)
        assert s == st_ref and tr.tolist() == tr_ref.tolist(), trial
        assert np.array_equal(_bits(Tg), _bits(Tr))


def test_resident_required_but_too_large(gpu):
    T = synth.raw_tableau(2049, 4100, seed=3)                  # 67 MB: does not fit 256 x 160 KB
    with gpu.DeviceTableau.from_host(T) as dt:
        with pytest.raises(gpu.LpxError, match="does not fit"):
            dt.primal_run(resident=1, max_iter=1)


def test_resident_falls_back_to_streaming_when_a_workgroup_is_missing(oracle):
    """LPX_RESIDENT_TEST_MUTE=1 makes one workgroup play dead: every exchange wait expires (bounded spins, ~2 s),
    nothing is written back, and an automatic run streams the same LP to the same bits; resident=1 reports it."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np, time
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        c, A, b = synth.dense_lp(40, 60, seed=2)
        T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = O.primal_tableau(Tr, br)
        with L.DeviceTableau.from_host(T, basis) as dt:
            dt.snapshot()
            try:
                dt.primal_run(resident=1)
                raise SystemExit("resident=1 should have failed")
            except L.LpxError as e:
                assert "exchange wait expired" in str(e), e
        with L.DeviceTableau.from_host(T, basis) as dt:
            t0 = time.time()
            status, st = dt.primal_run()
            Tg, bg = dt.download()
            assert status == st_ref and st["launches"] > 2
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
            assert dt.trace().tolist() == tr_ref.tolist()
            t1 = time.time()
            dt.upload(T, basis); dt.primal_run()          # the handle stays on the streaming path: no second wait
            assert time.time() - t1 < 0.5 * (t1 - t0) + 0.5
        print("OK")
    ''')
    env = dict(os.environ, LPX_RESIDENT_TEST_MUTE="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    # LPX_RESIDENT_TEST_MUTE=2: the workgroup dies from the SECOND launch on -- the streaming kernels take over in
    # mid-solve (pivot 32 of a chunked run) and the callbacks, trace and tableau still come out whole.
    code2 = textwrap.dedent('''
        import numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        c, A, b = synth.dense_lp(96, 150, seed=21)
        T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref = O.primal_tableau(Tr, br)
        assert len(tr_ref) > 40
        ev = []
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, st = dt.primal_run(resident=0, batch=16 * 2, cb=lambda it, r, q: ev.append((it, r, q)))
            Tg, bg = dt.download()
            assert status == st_ref and st["pivots"] == len(tr_ref)
            assert [e[0] for e in ev] == list(range(1, len(tr_ref) + 1))
            assert [(e[1], e[2]) for e in ev] == [tuple(x) for x in tr_ref.tolist()]
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
            assert dt.trace().tolist() == tr_ref.tolist()
        print("OK")
    ''')
    env["LPX_RESIDENT_TEST_MUTE"] = "2"
    r = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_resident_group_multi_run_matches_streaming_batch(oracle):
    """lpx_multi_run keeps a few node LPs resident at once (lpx_resident_group, blockIdx.y = node)
    and refills the slices between launches.  Mixed primal / dual nodes of different shapes must end bit-identical to
    the oracle -- and so to the default batched streaming run."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        g = np.random.default_rng(11)
        tabs, duals, refs = [], [], []
        for k in range(9):
            m, n = int(g.integers(20, 70)), int(g.integers(30, 90))
            c, A, b = synth.dense_lp(m, n, seed=100 + k)
            if k % 3 == 0:                                   # primal node
                T, basis = synth.primal_tableau_from(c, A, b)
                Tr, br = T.copy(), basis.copy()
                st, tr = O.primal_tableau(Tr, br)
                duals.append(0)
            else:                                            # dual node: negative right-hand sides, as a >= row leaves them
                b2 = b.copy(); b2[: 1 + k % 4] *= -0.05
                T, basis = synth.primal_tableau_from(c, A, b2)
                Tr, br = T.copy(), basis.copy()
                st, tr, nf = O.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
                duals.append(1)
            tabs.append((T, basis)); refs.append((st, Tr, br, tr))
        hs = [L.DeviceTableau.from_host(T, basis) for T, basis in tabs]
        statuses, stats = L.multi_run(hs, duals, dual_opts=L.default_opts(True, fdf_guard=10000, cleanup=1))
        for h, (st, Tr, br, tr), s, ss in zip(hs, refs, statuses, stats):
            Tg, bg = h.download()
            assert s == st, (s, st)
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
            assert h.trace().tolist() == tr.tolist() and ss["pivots"] == len(tr)
        assert max(ss["launches"] for ss in stats) < 50      # launches of the group kernel, not 2 per pivot
        print("OK")
    ''')
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("cleanup,guard", [(0, 100), (1, 10000)])
def test_dual_resident_and_streaming_paths_agree_with_the_oracle(gpu, oracle, cleanup, guard):
    """lpx_dual_run picks the resident group kernel by itself; the streaming dual kernels stay the path for large
    tableaux and for B&B batches -- both must give the oracle's bits (faithful guard 100 and repaired mode)."""
    g = np.random.default_rng(17)
    for trial in range(6):
        m, n = int(g.integers(12, 90)), int(g.integers(20, 120))
        c, A, b = synth.dense_lp(m, n, seed=300 + trial)
        b = b.copy(); b[: 1 + trial % 5] *= -0.04                  # negative right-hand sides, as a >= row leaves them
        if trial % 2:
            c = -c                                                 # dual feasible from the start: straight into the dual loop
        T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy()
        st_ref, tr_ref, nf = oracle.dual_tableau(Tr, br, fdf_guard=guard, cleanup=cleanup)
        for res in (1, -1):
            with gpu.DeviceTableau.from_host(T, basis) as dt:
                status, st = dt.dual_run(fdf_guard=guard, cleanup=cleanup, resident=res)
                Tg, bg = dt.download()
                assert status == st_ref, (trial, res)
                assert dt.trace().tolist() == tr_ref.tolist() and bg.tolist() == br.tolist()
                assert np.array_equal(_bits(Tg), _bits(Tr))
                assert st["fdf_pivots"] == nf and (st["launches"] <= 2) == (res == 1)


@pytest.mark.parametrize("mute", ["1", "2"])
def test_group_kernel_hands_over_to_the_streaming_kernels(oracle, mute):
    """LPX_RESIDENT_TEST_MUTE makes one workgroup of the group kernel play dead (1: from the first launch, 2: from the
    second launch on).  lpx_multi_run must finish every unfinished node on the batched streaming kernels from the pivot
    it had reached, lpx_dual_run likewise -- same bits as the oracle, callbacks complete and in order."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        tabs, refs = [], []
        for k in range(7):                                   # more nodes than slices, each longer than one 96-pivot launch
            c, A, b = synth.dense_lp(220 + 9 * k, 330 + 7 * k, seed=500 + k)
            A2, b2 = A.copy(), b.copy()
            A2[: 2 + k % 3] *= -1.0; b2[: 2 + k % 3] *= -0.05       # a few >= rows (feasible): negative right-hand sides
            T, basis = synth.primal_tableau_from(c, A2, b2)
            Tr, br = T.copy(), basis.copy()
            st, tr, nf = O.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
            assert len(tr) > 110
            tabs.append((T, basis)); refs.append((st, Tr, br, tr))
        hs = [L.DeviceTableau.from_host(T, basis) for T, basis in tabs]
        o = L.default_opts(True, fdf_guard=10000, cleanup=1)
        statuses, stats = L.multi_run(hs, [1] * len(hs), dual_opts=o)
        for h, (st, Tr, br, tr), s, ss in zip(hs, refs, statuses, stats):
            Tg, bg = h.download()
            assert s == st and ss["pivots"] == len(tr), (s, st, ss["pivots"], len(tr))
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
            assert h.trace().tolist() == tr.tolist()
        # single LP with callbacks, 32 pivots per launch
        T, basis = tabs[3]; st, Tr, br, tr = refs[3]
        ev = []
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, s1 = dt.dual_run(L.default_opts(True, fdf_guard=10000, cleanup=1, batch=32), cb=lambda it, r, q: ev.append((it, r, q)))
            Tg, bg = dt.download()
            assert status == st and s1["pivots"] == len(tr)
            assert [e[0] for e in ev] == list(range(1, len(tr) + 1)) and [(e[1], e[2]) for e in ev] == [tuple(x) for x in tr.tolist()]
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
        print("OK")
    ''')
    env = dict(os.environ, LPX_RESIDENT_TEST_MUTE=mute, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_late_workgroup_cannot_leave_a_half_pivoted_tableau(oracle):
    """LPX_RESIDENT_TEST_MUTE=3: the last workgroup gets its CU only AFTER the others have given up (the real
    non-co-residency case).  It still finds their first ratios, owns the pivot row, completes the only pivot of a
    max_iter=1 launch and stores its rows -- a pivot the other workgroups never made.  The host must put the tableau
    of the launch's start back before the streaming kernels take over (lpx_tableau.cpp run_resident /
    run_resident_group): result bit-identical to the oracle, not a tableau with one row pivoted twice."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        m, n = 8, 12
        for seed in range(1, 400):                            # an LP whose first leaving row belongs to the LAST workgroup
            c, A, b = synth.dense_lp(m, n, seed=seed)
            T, basis = synth.primal_tableau_from(c, A, b)
            Tr, br = T.copy(), basis.copy()
            st, tr = O.primal_tableau(Tr, br, max_iter=1)
            if len(tr) == 1 and tr[0][0] == m - 1:
                break
        else:
            raise SystemExit("no suitable seed")
        # primal resident kernel
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, s1 = dt.primal_run(max_iter=1)
            Tg, bg = dt.download()
            assert status == st == 3 and s1["pivots"] == 1 and dt.trace().tolist() == tr.tolist()
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
        # resident required: the launch is lost, the call fails, and the tableau is the one that went in
        with L.DeviceTableau.from_host(T, basis) as dt:
            try:
                dt.primal_run(max_iter=1, resident=1)
                raise SystemExit("expected an error")
            except L.LpxError as e:
                assert e.code == L._lib.EDEVICE
            Tg, bg = dt.download()
            assert np.array_equal(Tg.view(np.uint64), T.view(np.uint64)) and bg.tolist() == basis.tolist()
        # group kernel (one primal node through lpx_multi_run)
        with L.DeviceTableau.from_host(T, basis) as dt:
            statuses, stats = L.multi_run([dt], [0], primal_opts=L.default_opts(False, max_iter=1))
            Tg, bg = dt.download()
            assert statuses[0] == 3 and stats[0]["pivots"] == 1, (statuses, stats)
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist()
        print("OK")
    ''')
    env = dict(os.environ, LPX_RESIDENT_TEST_MUTE="3", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_column_owning_resident_kernel_is_bit_identical(oracle):
    """LPX_RESIDENT_COL=1: lpx_resident_primal_col (workgroups own columns, candidates + candidate columns in one exchange)
    gives the oracle's bits on ragged shapes, ties, every terminal status, chunked callbacks and config 2 in full."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np, hashlib
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        from oracle import oracle as O
        def bits(a): return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
        for (m, n, seed) in [(1, 1, 1), (3, 5, 2), (40, 64, 3), (257, 769, 4), (513, 1100, 5), (300, 7, 6), (1024, 2048, synth.SEED)]:
            c, A, b = synth.dense_lp(m, n, seed=seed)
            T, basis = synth.primal_tableau_from(c, A, b)
            Tr, br = T.copy(), basis.copy()
            st, tr = O.primal_tableau(Tr, br)
            ev = []
            with L.DeviceTableau.from_host(T, basis) as dt:
                status, s1 = dt.primal_run(L.default_opts(False, resident=1, batch=37 if m < 300 else 0),
                                           cb=(lambda it, r, q: ev.append((r, q))) if m < 300 else None)
                Tg, bg = dt.download()
                assert status == st and s1["pivots"] == len(tr), (m, n)
                assert dt.trace().tolist() == tr.tolist() and bg.tolist() == br.tolist()
                assert np.array_equal(bits(Tg), bits(Tr)), (m, n)
                if m < 300: assert [list(e) for e in ev] == tr.tolist()
        # degenerate ties and an unbounded column
        g = np.random.default_rng(5)
        A = g.integers(0, 3, size=(700, 9)).astype(float); b = g.integers(0, 2, size=700).astype(float)
        c = g.integers(1, 9, size=9).astype(float)
        T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy(); st, tr = O.primal_tableau(Tr, br)
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, _ = dt.primal_run(resident=1); Tg, bg = dt.download()
            assert status == st and dt.trace().tolist() == tr.tolist() and np.array_equal(bits(Tg), bits(Tr))
        T = np.array([[-1.0, 1.0, 1.0, 0.0, 1.0], [-2.0, 0.5, 0.0, 1.0, 3.0], [-1.0, -1.0, 0.0, 0.0, 0.0]]); basis = np.array([2, 3], np.int32)
        Tr, br = T.copy(), basis.copy(); st, tr = O.primal_tableau(Tr, br)
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, _ = dt.primal_run(resident=1)
            assert status == st == 1 and dt.trace().tolist() == tr.tolist()
        # iteration limit in the middle of a launch
        c, A, b = synth.dense_lp(64, 100, seed=9); T, basis = synth.primal_tableau_from(c, A, b)
        Tr, br = T.copy(), basis.copy(); st, tr = O.primal_tableau(Tr, br, max_iter=11)
        with L.DeviceTableau.from_host(T, basis) as dt:
            status, _ = dt.primal_run(resident=1, max_iter=11); Tg, bg = dt.download()
            assert status == st == 3 and dt.trace().tolist() == tr.tolist() and np.array_equal(bits(Tg), bits(Tr))
        print("OK")
    ''')
    env = dict(os.environ, LPX_RESIDENT_COL="1", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


def test_register_group_forms_agree_bit_for_bit(oracle):
    """lpx_resident_group_r in its forms -- three columns per lane with rows in the LDS beside the registers (default), without the LDS
    rows, two columns per lane -- against the LDS-resident group kernel and the streaming kernels: the same config-4 node LPs (root
    shape 770x1282 and deeper levels: the tile's padding columns differ), final tableaux, bases, pivot counts and statuses bitwise.
    The environment switches are read once per process, hence one child process per form."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import hashlib, numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        c, A, rel, b = synth.binary_ip(512, 256)
        n = len(c)
        h = hashlib.sha256()
        hs = []
        for depth in (1, 2, 3, 5):
            rows = [-np.eye(n)[k] if k % 2 == 0 else np.eye(n)[k] for k in range(depth)]      # x_k >= 1 as -x_k <= -1, x_k <= 0
            rhs = [-1.0 if k % 2 == 0 else 0.0 for k in range(depth)]
            T, basis = synth.primal_tableau_from(c, np.vstack([A] + rows), np.concatenate([b, rhs]))
            for copy in range(4):                                      # 16 nodes: more than the LDS form holds at a time
                hs.append(L.DeviceTableau.from_host(T, basis))
        o = L.default_opts(True, fdf_guard=10000, cleanup=1)
        po = L.default_opts(False)
        st, stats = L.multi_run(hs, [1] * len(hs), po, o)
        for dt, s, k in zip(hs, st, stats):
            Tg, bg = dt.download()
            h.update(np.ascontiguousarray(Tg).view(np.uint8)); h.update(np.ascontiguousarray(bg).view(np.uint8))
            h.update(np.array([s, k["pivots"]], dtype=np.int64).view(np.uint8))
        print("DIGEST", h.hexdigest(), sum(k["pivots"] for k in stats))
    ''')
    base = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    forms = {"default": {}, "no LDS rows": {"LPX_RESIDENT_REGS_LDS": "0"}, "two columns": {"LPX_RESIDENT_REGS_KC": "2"},
             "LDS form": {"LPX_RESIDENT_REGS": "0"}, "streaming": {"LPX_RESIDENT_GROUP": "0"}}
    seen = {}
    for name, extra in forms.items():
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "DIGEST" in r.stdout, name + ": " + r.stdout + r.stderr
        seen[name] = r.stdout.split("DIGEST", 1)[1].split()
    assert len({tuple(v) for v in seen.values()}) == 1, seen
    assert int(seen["default"][1]) > 1000, seen


@pytest.mark.parametrize("dual", [0, 1])
def test_group_of_more_nodes_than_compute_units(gpu, oracle, dual):
    """lpx_multi_run with 300 small LPs (more nodes than the chip has compute units: the planner divided by zero there before r03),
    primal loop and dual loop: every one ends as the oracle says, bit for bit."""
    hs, want = [], []
    for k in range(300):
        m, n = 3 + k % 5, 4 + k % 7
        c, A, b = synth.dense_lp(m, n, seed=1000 + k)
        T, basis = synth.primal_tableau_from(c, A, b)
        if dual:                                                       # one repaired >= row: -row <= -b'
            T[k % m, :n] *= -1.0
            T[k % m, -1] = -0.02 * T[k % m, -1]
        Tr, br = T.copy(), basis.copy()
        if dual:
            st, tr, _ = oracle.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
        else:
            st, tr = oracle.primal_tableau(Tr, br)
        want.append((st, len(tr), Tr, br))
        hs.append(gpu.DeviceTableau.from_host(T, basis))
    st, stats = gpu.multi_run(hs, [dual] * len(hs), None, gpu.default_opts(True, fdf_guard=10000, cleanup=1))
    for dt, s, k, (ws, wp, Tr, br) in zip(hs, st, stats, want):
        Tg, bg = dt.download()
        assert s == ws and k["pivots"] == wp
        assert bg.tolist() == br.tolist() and np.array_equal(_bits(Tg), _bits(Tr))
        dt.close()
