"""GPU parity at the reference's plugin boundary: LPSolver / ILPAlgorithm mirrors (C++ host -> C ABI ->
HIP) against the oracle and the hand-derived KATs.  Reads like tests of the reference would."""
import json
import os

import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


def _oracle_problem(oracle, p):
    A = np.array([c.A for c in p.Constraints], dtype=float)
    return oracle.Problem(int(p.ObjectiveSense), p.C, A, [int(c.Relation) for c in p.Constraints],
                          [c.B for c in p.Constraints])


def test_kat1_primal_through_lpsolver(gpu):
    k = KATS["kat1_primal"]
    p = gpu.ParseFromText(k["text"])
    texts = []
    r = gpu.LPSolver().Solve(p, "  primal   simplex ALGORITHM ", lambda t, h: texts.append((t, h)))
    assert r.Status == 0 and r.OptimalValue == k["z"]
    assert r.Solution.tolist() == k["x"] and r.Basis.tolist() == k["basis"] and r.Trace.tolist() == k["trace"]
    assert r.VarNames == ["x1", "x2", "c1", "c2", "c3"]
    assert "Status: OPTIMAL" in r.Report and "  x1 = 2\n  x2 = 6\n  z* = 36\n" in r.Report
    assert r.Report.startswith("Objective: max +3x1 +5x2\nSubject to:\n  +1x1 +0x2 <= 4\n")
    assert r.Summary == "Status: OPTIMAL\nz* = 36\nx* = [2, 6]\n"
    assert texts[0][0].startswith("TABLEAU Iteration 0\n       Basis          x1          x2          c1")
    assert len(texts) == 3 and texts[1][1] is None


def test_render_iterations_matches_reference_format(gpu):
    p = gpu.ParseFromText(KATS["kat1_primal"]["text"])
    texts = []
    gpu.LPSolver(render_iterations=1).Solve(p, "Primal Simplex", lambda t, h: texts.append((t, h)))
    assert len(texts) == 3
    t1, h1 = texts[1]
    lines = t1.split("\n")
    assert lines[0] == "TABLEAU Iteration 1"
    assert lines[1] == "".join(s.rjust(12) for s in ["Basis", "x1", "x2", "c1", "c2", "c3", "RHS"])
    assert lines[2] == "-" * (12 * 7)
    assert lines[3] == "".join(s.rjust(12) for s in ["z", "-3", "0", "0", "2.5", "0", "30"])
    assert lines[5] == "".join(s.rjust(12) for s in ["x2", "0", "1", "0", "0.5", "0", "6"])
    assert h1.shape == (4, 6) and h1[1].all() and h1[:, 1].all() and h1.sum() == 6 + 4 - 1


def test_render_iterations_revised_matches_reference_block(gpu):
    """BuildIterationBlock / MatrixToString (Models/RevisedPrimalSimplex.cs:191-261) on KAT-1, derived by
    hand from the cited lines -- including the reference's quirks: r_N is labelled with the Nidx of AFTER
    the pivot (:219-220) and the ratio lines divide the x_B of after it (:131,:139,:236)."""
    p = gpu.ParseFromText(KATS["kat1_primal"]["text"])
    texts = []
    gpu.LPSolver(render_iterations=1).Solve(p, "Revised Primal Simplex", lambda t, h: texts.append((t, h)))
    assert len(texts) == 3
    cell = lambda *v: "".join(str(x).rjust(12) for x in v) + "\n"
    assert texts[0][1] is None and texts[0][0] == (
        "=== Revised Simplex Iteration 0 ===\nBasis: c1, c2, c3\nNonbasic: x1, x2\n\n"
        "Product-form: current B^{-1}\n" + cell(1, 0, 0) + cell(0, 1, 0) + cell(0, 0, 1) +
        "x_B = [4, 12, 18]\nz = 0\n\n")
    t1, h1 = texts[1]
    assert t1 == (
        "=== Revised Simplex Iteration 1 ===\nBasis: c1, x2, c3\nNonbasic: x1, c2\n\n"
        "Product-form: current B^{-1}\n" + cell(1, 0, 0) + cell(0, "0.5", 0) + cell(0, -1, 1) +
        "x_B = [4, 6, 6]\nz = -30\n\n"
        "Reduced costs (r_N = c_N - c_B^T B^{-1} N):\n  r(0:x1) = -3\n  r(3:c2) = -5\n\n"
        "Entering variable: x2\nDirection d = B^{-1} * a_entering:\n  d = [0, 2, 2]\n\n"
        "Ratio test (theta):\n  row 1: d_i <= 0 (skip)\n  row 2: 6 / 2 = 3\n  row 3: 6 / 2 = 3\n"
        "Chosen theta* = 6\n\n")
    assert h1.shape == (3, 4) and h1[1].all() and h1.sum() == 4
    t2, _ = texts[2]
    assert "Basis: c1, x2, x1\nNonbasic: c2, c3\n" in t2 and "x_B = [2, 6, 2]\nz = -36\n" in t2
    assert "  r(3:c2) = -3\n  r(4:c3) = 2.5\n" in t2 and "Chosen theta* = 2\n" in t2


def test_kat3_min_sign(gpu):
    k = KATS["kat3_min_sign"]
    r = gpu.PrimalSimplex().Solve(gpu.ParseFromText(k["text"]))
    assert r.OptimalValue == k["z"] and r.Solution.tolist() == k["x"]


def test_kat5_eq(gpu):
    k = KATS["kat5_eq_primal"]
    r = gpu.PrimalSimplex().Solve(gpu.ParseFromText(k["text"]))
    assert r.Tableau.tolist() == k["final_tableau"] and r.Basis.tolist() == k["basis"] and r.OptimalValue == k["z"]


def test_exceptions_carry_reference_messages(gpu):
    P, C_, S, R = gpu.LPProblem, gpu.Constraint, gpu.Sense, gpu.Rel
    with pytest.raises(gpu.SolverException, match="Constraint contains '>=' sign") as e:
        gpu.PrimalSimplex().Solve(P(S.Max, [1, 1], [C_([1, 1], R.GE, 1)]))
    assert e.value.code == gpu._lib.E_GE_PRESENT
    with pytest.raises(gpu.SolverException, match="negative RHS value"):
        gpu.PrimalSimplex().Solve(P(S.Max, [1, 1], [C_([1, 1], R.LE, -1)]))
    with pytest.raises(gpu.SolverException, match="supports only <= constraints"):
        gpu.RevisedPrimalSimplex().Solve(P(S.Max, [1], [C_([1], R.EQ, 1)]))
    with pytest.raises(gpu.SolverException, match="Algorithm not supported: 'x'"):
        gpu.LPSolver().Solve(P(S.Max, [1], [C_([1], R.LE, 1)]), "x")
    with pytest.raises(gpu.SolverException, match="No algorithm selected"):
        gpu.LPSolver().Solve(P(S.Max, [1], [C_([1], R.LE, 1)]), "   ")
    with pytest.raises(gpu.SolverException, match="exactly one constraint"):
        gpu.BranchAndBoundKnapsack().Solve(P(S.Max, [1, 2], [C_([1, 1], R.LE, 1), C_([1, 1], R.LE, 2)]))
    with pytest.raises(gpu.SolverException, match="Iteration limit exceeded"):
        c, A, b = synth.dense_lp(40, 60, seed=2)
        gpu.LPSolver(max_iter=5).Solve(P.from_arrays(0, c, A, np.zeros(40, int), b), "primal")
    r = gpu.PrimalSimplex().Solve(P(S.Max, [1, 0], [C_([-1, 1], R.LE, 1)]))
    assert r.Status == 1 and "UNBOUNDED" in r.Report and r.Solution is not None     # a status, not an exception


def test_kat7_dual_defects_and_repair(gpu):
    k = KATS["kat7_dual_d1"]
    p = gpu.ParseFromText(k["text"])
    f = gpu.DualSimplex().Solve(p)
    assert f.Solution is None and f.Tableau is None and f.Basis is None and f.VarNames is None   # D2
    assert f.OptimalValue == 0.0 and f.Trace.tolist() == k["faithful"]["trace"]
    assert "z* = 36" in f.Summary                      # D1: x1 >= 3 silently became x1 <= 3
    r = gpu.DualSimplex(dual_flags=7).Solve(p)
    assert r.Solution.tolist() == k["repaired"]["x"] and r.OptimalValue == k["repaired"]["z"]
    assert r.Trace.tolist() == k["repaired"]["trace"]


def test_kat2_revised_text_only(gpu):
    k = KATS["kat2_revised"]
    r = gpu.RevisedPrimalSimplex().Solve(gpu.ParseFromText(k["text"]))
    assert r.Solution is None and r.Tableau is None                      # text only, RevisedPrimalSimplex.cs:294
    assert r.Summary == "Status: OPTIMAL\nx* = [2, 6]\nz* = 36\n"
    assert r.Trace.tolist() == k["trace"] and r.Aux[0] == 36.0 and r.Aux[1] == -36.0


@pytest.mark.parametrize("mode,key", [(0, "faithful"), (1, "repaired")])
@pytest.mark.parametrize("search", [0, 1])
def test_kat6_bnb(gpu, mode, key, search):
    k = KATS["kat6_bnb"]
    r = gpu.BranchAndBound(bnb_mode=mode, bnb_search=search, concurrent_nodes=2).Solve(gpu.ParseFromText(k["text"]))
    e = k[key]
    assert r.OptimalValue == e["best_z"] and r.Solution.tolist() == e["best_x"]
    if search == 0:
        assert r.LpSolves == e["lp_solves"] and r.NodeLog.tolist() == e["log"]
    assert r.Report.startswith("Branch & Bound Finished.\nBest integer z* = %.3f\n" % e["best_z"])
    assert r.Tableau is not None and r.Tableau.shape == (3, 5)            # root tableau, Branch&Bound.cs:118-120


@pytest.mark.parametrize("mode", [0, 1])
def test_bnb_dfs_identical_to_oracle_on_random_binary_ips(gpu, oracle, mode):
    g = np.random.default_rng(17)
    for trial in range(4):
        n, m = 10, 5
        A = g.integers(0, 10, size=(m, n)).astype(float)
        b = np.floor(0.5 * A.sum(axis=1))
        c = g.integers(1, 21, size=n).astype(float)
        Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
        p = gpu.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
        ref = oracle.bnb_solve(_oracle_problem(oracle, p), mode)
        r = gpu.BranchAndBound(bnb_mode=mode).Solve(p)
        assert r.NodeLog.tolist() == ref.log.tolist(), trial
        assert r.LpSolves == ref.lp_solves and r.Nodes == ref.nodes_visited
        assert np.array_equal(r.NodeZ.view(np.uint64), ref.log_z.view(np.uint64))
        if ref.has_incumbent:
            assert r.OptimalValue == ref.best_z and r.Solution.tolist() == ref.best_x.tolist()
        assert r.Stats["pivots"] + 0 >= 0
        # the sharded level search must reach the same optimum
        if mode == 1:
            lv = gpu.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=4).Solve(p)
            assert lv.OptimalValue == ref.best_z


def test_knapsack_relax_batch_vs_oracle(gpu, oracle):
    g = np.random.default_rng(23)
    for n in (5, 37, 600, 5000):
        w = g.integers(1, 1001, size=n).astype(float)
        p = w + g.integers(0, 101, size=n)
        cap = float(np.floor(0.5 * w.sum()))
        dk = gpu.DeviceKnapsack(p, w, cap)
        order = oracle.knapsack_order(p, w)
        assert dk.order().tolist() == order.tolist()
        nodes = [{}]
        for t in range(40):
            k = int(g.integers(0, min(n, 30)))
            idx = g.choice(n, size=k, replace=False)
            nodes.append({int(i): int(g.integers(0, 2)) for i in idx})
        nodes.append({int(i): 1 for i in range(n)})          # everything fixed in: overflow (:455-456)
        nodes.append({int(i): 0 for i in range(n)})
        P, W, F, X = dk.relax_batch(nodes)
        for j, nd in enumerate(nodes):
            a = -np.ones(n, np.int32)
            for i, v in nd.items():
                a[i] = v
            rp, rw, rf, rx = oracle.knapsack_relax(p, w, cap, order, a, want_vector=True)
            assert (P[j], W[j], F[j]) == (rp, rw, rf), (n, j)
            if rf >= 0:
                assert X[j] == rx[order[rf]]
        dk.close()


def test_knapsack_relax_batch2_children_vs_oracle(gpu, oracle):
    """lpx_knapsack_relax_batch2: slot 0 = the node, slots 1 / 2 = the node with its fractional item fixed to 0 / 1
    (what the best-first loop asks for when it pops the node, BranchAndBoundKnapsack.cs:180,207-209,267-269)."""
    g = np.random.default_rng(31)
    for n in (6, 41, 700, 5000):
        w = g.integers(1, 1001, size=n).astype(float)
        p = w + g.integers(0, 101, size=n)
        cap = float(np.floor(0.5 * w.sum()))
        dk = gpu.DeviceKnapsack(p, w, cap)
        order = oracle.knapsack_order(p, w)
        nodes = [{}]
        for t in range(40):
            k = int(g.integers(0, min(n, 300)))              # deeper than the 256 register-cached entries too
            idx = g.choice(n, size=k, replace=False)
            nodes.append({int(i): int(g.integers(0, 2)) for i in idx})
        nodes.append({int(i): 1 for i in range(n)})          # overflow: no children
        nodes.append({int(i): 0 for i in range(n)})          # nothing left: no fractional item
        P, W, F, X = dk.relax_batch2(nodes)
        P1, W1, F1, X1 = dk.relax_batch(nodes)
        assert (P[:, 0].tolist(), W[:, 0].tolist(), F[:, 0].tolist(), X[:, 0].tolist()) == (P1.tolist(), W1.tolist(), F1.tolist(), X1.tolist())
        seen_children = 0
        for j, nd in enumerate(nodes):
            if F[j, 0] < 0:
                assert F[j, 1] == -2 and F[j, 2] == -2, (n, j)
                continue
            item = int(order[F[j, 0]])
            for v in (0, 1):
                a = -np.ones(n, np.int32)
                for i, val in nd.items():
                    a[i] = val
                a[item] = v
                rp, rw, rf, rx = oracle.knapsack_relax(p, w, cap, order, a, want_vector=True)
                assert (P[j, 1 + v], W[j, 1 + v], F[j, 1 + v]) == (rp, rw, rf), (n, j, v)
                if rf >= 0:
                    assert X[j, 1 + v] == rx[order[rf]]
                seen_children += 1
        assert seen_children >= 40
        dk.close()


def test_kat8_knapsack_and_oracle_parity(gpu, oracle):
    k = KATS["kat8_knapsack"]
    P, C_, S, R = gpu.LPProblem, gpu.Constraint, gpu.Sense, gpu.Rel
    r = gpu.BranchAndBoundKnapsack().Solve(P(S.Max, k["profit"], [C_(k["weight"], R.LE, k["cap"])]))
    assert r.OptimalValue == k["best_z"] and r.Extra.tolist() == k["best_x"] and r.Nodes == k["nodes_popped"]
    assert "Status: BEST CANDIDATE FOUND" in r.Report and "Best Candidate = 220.5" in r.Report and r.Summary == ""
    g = np.random.default_rng(4)
    for n, cap_nodes in ((18, 0), (60, 0), (400, 3000)):
        w = g.integers(1, 60, size=n).astype(float)
        p = w + g.integers(0, 12, size=n)
        cap = float(np.floor(0.5 * w.sum()))
        ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=cap_nodes)
        r = gpu.BranchAndBoundKnapsack(max_nodes=cap_nodes).Solve(P(S.Max, p.tolist(), [C_(w.tolist(), R.LE, cap)]))
        assert r.OptimalValue == ref.best_z, n
        assert r.Nodes == ref.nodes_popped and r.Aux[0] == ref.relaxations and r.Aux[2] == ref.nodes_expanded
        assert r.Extra.astype(int).tolist() == ref.best_x.tolist()


def test_knapsack_search_is_independent_of_the_speculation_width(gpu, oracle):
    """The evaluated-leaves heap decides only WHEN a relaxation is computed, never what the search does with it: popped /
    relaxations / expanded / largest heap / z / x equal the oracle's for every number of leaves expanded ahead per launch
    (2 = nearly on demand ... 4096 = far ahead), with a node budget and to exhaustion."""
    P, C_, S, R = gpu.LPProblem, gpu.Constraint, gpu.Sense, gpu.Rel
    g = np.random.default_rng(91)
    for n, cap_nodes in ((300, 0), (3000, 5000), (20000, 7000)):
        w = g.integers(1, 1001, size=n).astype(float)
        p = w + g.integers(0, 101, size=n)
        cap = float(np.floor(0.5 * w.sum()))
        ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=cap_nodes)
        launches = []
        for width in (2, 37, 256, 4096):
            r = gpu.BranchAndBoundKnapsack(max_nodes=cap_nodes, concurrent_nodes=width).Solve(P(S.Max, p.tolist(), [C_(w.tolist(), R.LE, cap)]))
            assert r.Nodes == ref.nodes_popped and r.Aux[0] == ref.relaxations and r.Aux[2] == ref.nodes_expanded, (n, width)
            assert r.Aux[3] == ref.max_heap, (n, width)
            if np.isfinite(ref.best_z):
                assert r.OptimalValue == ref.best_z and r.Extra.astype(int).tolist() == ref.best_x.tolist(), (n, width)
            launches.append(r.Stats["launches"])
        assert launches[0] >= launches[2] >= launches[3]      # wider speculation = fewer, larger launches
        if n >= 3000:
            assert launches[0] > 2 * launches[2]


@pytest.mark.parametrize("mode,key", [(0, "faithful"), (1, "repaired")])
def test_kat9_bnb_revised(gpu, mode, key):
    """BranchAndBoundRevised -- node x*, z* re-parsed from the 3-decimal Summary text."""
    k = KATS["kat9_bnb_revised"]
    r = gpu.BranchAndBoundRevised(bnb_mode=mode).Solve(gpu.ParseFromText(k["text"]))
    e = k[key]
    assert r.OptimalValue == e["best_z"] and r.Extra.tolist() == e["best_x"] and r.LpSolves == e["lp_solves"]
    assert r.NodeLog.tolist() == e["log"] and r.NodeZ.tolist() == e["log_z"]
    assert r.Solution is None and r.Tableau is None        # text only, BranchAndBoundRevised.cs:92-96
    assert r.Report == r.Summary and r.Report.startswith("Revised Branch & Bound Finished.\nBest integer z* = %g\n" % e["best_z"])


@pytest.mark.parametrize("mode", [0, 1])
def test_bnb_revised_identical_to_oracle_on_random_binary_ips(gpu, oracle, mode):
    g = np.random.default_rng(29)
    for trial in range(3):
        n, m = 8, 4
        A = g.integers(0, 10, size=(m, n)).astype(float)
        b = np.floor(0.5 * A.sum(axis=1))
        c = g.integers(1, 21, size=n).astype(float)
        Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
        p = gpu.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
        ref = oracle.bnb_solve(_oracle_problem(oracle, p), mode, revised=True)
        r = gpu.BranchAndBoundRevised(bnb_mode=mode).Solve(p)
        assert r.NodeLog.tolist() == ref.log.tolist(), trial
        assert r.NodeZ.tolist() == ref.log_z.tolist() and r.LpSolves == ref.lp_solves
        if ref.has_incumbent:
            assert r.OptimalValue == ref.best_z and r.Extra.tolist() == ref.best_x.tolist()


def test_bnb_warm_started_children_reach_the_same_optimum(gpu, oracle):
    """SURVEY 8f rank 3 (bnb_search=2): children start from the parent's final tableau + the branching row and
    run the dual loop only.  Same LP optimum per node -> same B&B optimum as the reference-order search; node
    z of the root's two children must equal the cold re-solve to 1e-9 (they are the same LPs)."""
    g = np.random.default_rng(41)
    for trial in range(5):
        n, m = 12, 5
        A = g.integers(0, 10, size=(m, n)).astype(float)
        b = np.floor(0.5 * A.sum(axis=1))
        c = g.integers(1, 21, size=n).astype(float)
        Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
        p = gpu.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
        ref = oracle.bnb_solve(_oracle_problem(oracle, p), 1)
        cold = gpu.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=4).Solve(p)
        warm = gpu.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=4).Solve(p)
        assert cold.OptimalValue == ref.best_z, trial
        assert abs(warm.OptimalValue - ref.best_z) <= 1e-9 * abs(ref.best_z), trial     # different pivot path, same optimum
        assert abs(np.asarray(warm.Solution) @ c - ref.best_z) <= 1e-9 * abs(ref.best_z)
        # level order: entry 0 = root, entries 1,2 = its ceil / floor children in both searches
        if len(cold.NodeZ) >= 3 and len(warm.NodeZ) >= 3:
            for a, bz in zip(cold.NodeZ[:3], warm.NodeZ[:3]):
                assert abs(a - bz) <= 1e-9 * max(1.0, abs(a))
            assert cold.NodeLog[:3].tolist() == warm.NodeLog[:3].tolist()
        # and it needs far fewer pivots
        assert warm.Stats["pivots"] <= cold.Stats["pivots"]


@pytest.mark.parametrize("search", [1, 2])
def test_bnb_dive_policy_reaches_the_same_optimum(gpu, oracle, search):
    """bnb_dive=1: each round takes the deepest K nodes of the pool (depth-first-K) -- same optimum."""
    g = np.random.default_rng(43)
    for trial in range(4):
        n, m = 12, 5
        A = g.integers(0, 10, size=(m, n)).astype(float)
        b = np.floor(0.5 * A.sum(axis=1))
        c = g.integers(1, 21, size=n).astype(float)
        Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
        p = gpu.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
        ref = oracle.bnb_solve(_oracle_problem(oracle, p), 1)
        r = gpu.BranchAndBound(bnb_mode=1, bnb_search=search, bnb_dive=1, concurrent_nodes=3).Solve(p)
        assert abs(r.OptimalValue - ref.best_z) <= 1e-9 * abs(ref.best_z), (search, trial)
        assert abs(np.asarray(r.Solution) @ c - ref.best_z) <= 1e-9 * abs(ref.best_z)


@pytest.mark.parametrize("wide", ["1", "0"])
def test_knapsack_device_node_store_expand_batch(gpu, oracle, monkeypatch, wide):
    """lpx_knapsack_expand_batch: nodes are derived on the device from a stored parent plus one decision; the stored lists
    (returned in ascending item index) and the three relaxations per job equal what lpx_knapsack_relax_batch2 / the oracle
    give for the same fixed sets -- random chains 560 deep: the wide kernel (64 probes per round, lists <= 512 decisions)
    hands over to the one-probe-per-step kernel beyond its capacity; LPX_KNAP_WIDE=0 runs the latter throughout."""
    monkeypatch.setenv("LPX_KNAP_WIDE", wide)
    g = np.random.default_rng(77)
    for n in (7, 60, 900, 5000):
        w = g.integers(1, 1001, size=n).astype(float)
        p = w + g.integers(0, 101, size=n)
        cap = float(np.floor(0.5 * w.sum()))
        dk = gpu.DeviceKnapsack(p, w, cap)
        order = oracle.knapsack_order(p, w)
        known = {-1: {}}                                   # id -> fixed dict
        frontier = [-1]
        for gen in range(min(n - 1, 560 if wide == "1" else 300)):
            parents, items, vals = [], [], []
            for par in frontier[:6]:
                free = [i for i in g.choice(n, size=min(n, 8), replace=False) if int(i) not in known[par]]
                if not free:
                    continue
                parents.append(par); items.append(int(free[0])); vals.append(int(g.integers(0, 4) == 0))
            if not parents:
                break
            ids, P, W, F, X = dk.expand_batch(parents, items, vals)
            nodes = []
            for j, (par, it, v) in enumerate(zip(parents, items, vals)):
                fx = dict(known[par]); fx[it] = v
                known[int(ids[j])] = fx
                nodes.append(fx)
                if gen % 40 == 0 or gen > 540 or 505 < gen < 520:
                    assert dk.node_list(int(ids[j])) == fx, (n, gen, j)
                if F[j, 0] >= 0:                           # stored children: node + fractional item fixed to 0 / 1
                    item2 = int(order[F[j, 0]])
                    for c in (0, 1):
                        fx2 = dict(fx); fx2[item2] = c
                        known[int(ids[j]) + 1 + c] = fx2
                        if gen % 40 == 0:
                            assert dk.node_list(int(ids[j]) + 1 + c) == fx2
            P2, W2, F2, X2 = dk.relax_batch2(nodes)
            live = F != -2                                  # slots of children that do not exist carry no numbers
            assert np.array_equal(F, F2) and np.array_equal(P[live], P2[live]) and np.array_equal(W[live], W2[live]), (n, gen)
            assert np.array_equal(X[live], X2[live]), (n, gen)
            if gen % 25 == 0:
                for j, nd in enumerate(nodes):
                    a = -np.ones(n, np.int32)
                    for i, v in nd.items():
                        a[i] = v
                    rp, rw, rf, rx = oracle.knapsack_relax(p, w, cap, order, a, want_vector=True)
                    assert (P[j, 0], W[j, 0], F[j, 0]) == (rp, rw, rf), (n, gen, j)
            # next generation: mix of plain children and stored depth-2 children
            nxt = [int(i) for i in ids]
            nxt += [int(ids[j]) + 1 + int(g.integers(0, 2)) for j in range(len(ids)) if F[j, 0] >= 0][:3]
            frontier = nxt
        dk.close()


def test_knapsack_solve_same_counts_with_and_without_the_device_store(oracle):
    """LPX_KNAP_STORE=0 (every job ships its list, the r01 path) and the device-resident store give the same search."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import numpy as np, json
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        p, w, cap = synth.knapsack(20000, seed=5)
        kp = L.LPProblem(L.Sense.Max, p.tolist(), [L.Constraint(w.tolist(), L.Rel.LE, cap)])
        r = L.BranchAndBoundKnapsack(max_nodes=6000).Solve(kp)
        print(json.dumps([r.Nodes, r.Aux[0], r.Aux[2], r.Aux[3], r.OptimalValue if np.isfinite(r.OptimalValue) else None,
                          r.Extra.astype(int).tolist() if np.isfinite(r.OptimalValue) else None]))
    ''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for store in ("1", "0"):
        env = dict(os.environ, LPX_KNAP_STORE=store, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
    p, w, cap = synth.knapsack(20000, seed=5)
    ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=6000)
    import json
    got = json.loads(outs[0])
    assert got[0] == ref.nodes_popped and got[1] == ref.relaxations and got[2] == ref.nodes_expanded


def test_batched_solution_readback_and_parking_equal_the_per_node_calls(gpu):
    """lpx_multi_solution / lpx_store_save_multi (the B&B group's read-back and parent parking in one launch / one wait)
    against lpx_tableau_solution2 / lpx_store_save node by node: x, z, basis equal; a child built from a slot parked by
    either call is the same tableau, bit for bit."""
    import ctypes as C
    lib = gpu._lib.lib()
    g = np.random.default_rng(5)
    m, n, count = 37, 21, 6
    R, Cc = m + 1, n + m + 1
    ts = []
    for k in range(count):
        A = g.random((m, n)); b = 0.5 * n * g.uniform(0.9, 1.1, m); c = g.uniform(0.5, 1.5, n)
        T, basis = synth.primal_tableau_from(c, A, b)
        t = gpu.DeviceTableau(R + 1, Cc + 1)            # capacity for one more row / column: the warm-start child
        gpu._lib.check(lib.lpx_tableau_set_shape(t._h, R, Cc))
        t.R, t.C = R, Cc
        t.upload(T, basis)
        status, _ = t.primal_run()
        assert status == 0
        ts.append(t)
    arr = (C.c_void_p * count)(*[t._h for t in ts])
    x = np.zeros((count, n)); z = np.zeros(count); bs = -np.ones((count, m + 3), np.int32)
    gpu._lib.check(lib.lpx_multi_solution(arr, count, n, x.ctypes.data_as(C.POINTER(C.c_double)), z.ctypes.data_as(C.POINTER(C.c_double)),
                                          bs.ctypes.data_as(C.POINTER(C.c_int32)), m + 3))
    for k, t in enumerate(ts):
        x1 = np.zeros(n); z1 = C.c_double(); b1 = np.zeros(m, np.int32)
        gpu._lib.check(lib.lpx_tableau_solution2(t._h, n, x1.ctypes.data_as(C.POINTER(C.c_double)), C.byref(z1), b1.ctypes.data_as(C.POINTER(C.c_int32))))
        assert np.array_equal(x[k].view(np.uint64), x1.view(np.uint64)) and z[k] == z1.value
        assert bs[k, :m].tolist() == b1.tolist() and bs[k, m:].tolist() == [-1, -1, -1]
    store = C.c_void_p()
    gpu._lib.check(lib.lpx_store_create(R + 1, Cc + 1, C.byref(store)))
    try:
        slots = (C.c_int * count)()
        stores = (C.c_void_p * count)(*[store] * count)
        gpu._lib.check(lib.lpx_store_save_multi(stores, arr, count, slots))
        assert len(set(slots)) == count
        singles = []
        for k, t in enumerate(ts):
            one = C.c_int(-1)
            gpu._lib.check(lib.lpx_store_save(store, t._h, C.byref(one)))
            assert one.value not in list(slots)
            # the branching variable: any basic structural variable of this node
            T0, b0 = t.download()
            row = next(i for i in range(m) if b0[i] < n)
            kids = []
            for slot in (slots[k], one.value):
                ch = gpu.DeviceTableau(R + 1, Cc + 1)
                gpu._lib.check(lib.lpx_tableau_build_child_from_store(ch._h, store, slot, int(b0[row]), row, 0, float(np.floor(T0[row, -1]))))
                ch.R, ch.C = R + 1, Cc + 1
                kids.append(ch.download()); ch.close()
            assert np.array_equal(kids[0][0].view(np.uint64), kids[1][0].view(np.uint64)) and kids[0][1].tolist() == kids[1][1].tolist()
            singles.append((kids[0], int(b0[row]), row, float(np.floor(T0[row, -1]))))
        # ... and the children of all nodes in ONE launch (lpx_tableau_build_children_from_store)
        chs = [gpu.DeviceTableau(R + 1, Cc + 1) for _ in ts]
        carr = (C.c_void_p * count)(*[t._h for t in chs])
        var = np.array([x[1] for x in singles], np.int32); rowv = np.array([x[2] for x in singles], np.int32)
        ge = np.zeros(count, np.int32); bd = np.array([x[3] for x in singles])
        gpu._lib.check(lib.lpx_tableau_build_children_from_store(carr, stores, slots, count, var.ctypes.data_as(gpu._lib.ip), rowv.ctypes.data_as(gpu._lib.ip),
                                                                 ge.ctypes.data_as(gpu._lib.ip), bd.ctypes.data_as(gpu._lib.dp)))
        for ch, (ref, _, _, _) in zip(chs, singles):
            ch.R, ch.C = R + 1, Cc + 1
            Tc, bc = ch.download(); ch.close()
            assert np.array_equal(Tc.view(np.uint64), ref[0].view(np.uint64)) and bc.tolist() == ref[1].tolist()
    finally:
        lib.lpx_store_destroy(store)
        for t in ts:
            t.close()


def test_knapsack_expand_in_two_halves_equals_the_batch_call(gpu):
    """lpx_knapsack_expand_begin / _finish (the host works while the device evaluates) give what lpx_knapsack_expand_batch
    gives; a second _begin while a batch is in flight is refused, _finish without a batch is a no-op."""
    import ctypes as C
    lib = gpu._lib.lib()
    g = np.random.default_rng(12)
    n = 4000
    w = g.integers(1, 1001, size=n).astype(float); p = w + g.integers(0, 101, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    a, b = gpu.DeviceKnapsack(p, w, cap), gpu.DeviceKnapsack(p, w, cap)
    try:
        parents = [-1] * 5; items = [int(i) for i in g.choice(n, 5, replace=False)]; vals = [0, 1, 0, 1, 1]
        ids_b, P, W, F, X = b.expand_batch(parents, items, vals)
        par = np.asarray(parents, np.int64); it = np.asarray(items, np.int32); vv = np.asarray(vals, np.int8)
        ids = np.zeros(5, np.int64)
        args = (a._h, 5, par.ctypes.data_as(C.POINTER(C.c_int64)), it.ctypes.data_as(gpu._lib.ip), vv.ctypes.data_as(C.POINTER(C.c_int8)),
                ids.ctypes.data_as(C.POINTER(C.c_int64)))
        assert lib.lpx_knapsack_expand_finish(a._h, None, None, None, None) == 0          # nothing in flight
        gpu._lib.check(lib.lpx_knapsack_expand_begin(*args))
        assert lib.lpx_knapsack_expand_begin(*args) < 0                                   # one batch per handle
        pp = np.zeros(15); ww = np.zeros(15); fr = np.zeros(15, np.int32); fx = np.zeros(15)
        gpu._lib.check(lib.lpx_knapsack_expand_finish(a._h, pp.ctypes.data_as(gpu._lib.dp), ww.ctypes.data_as(gpu._lib.dp),
                                                      fr.ctypes.data_as(gpu._lib.ip), fx.ctypes.data_as(gpu._lib.dp)))
        live = F.reshape(-1) != -2
        assert ids.tolist() == ids_b.tolist() and fr.tolist() == F.reshape(-1).tolist()
        assert np.array_equal(pp[live], P.reshape(-1)[live]) and np.array_equal(ww[live], W.reshape(-1)[live]) and np.array_equal(fx[live], X.reshape(-1)[live])
        for j in range(5):
            assert a.node_list(int(ids[j])) == b.node_list(int(ids_b[j])) == {items[j]: vals[j]}
    finally:
        a.close(); b.close()


def test_batched_node_assembly_equals_the_per_node_call(gpu):
    """lpx_tableau_build_nodes (one launch for a group of B&B nodes) against lpx_tableau_build_node node by node: tableau,
    basis and live shape identical, for nodes of different depths (0 .. 5 branching rows) in handles of one capacity class."""
    import ctypes as C
    lib = gpu._lib.lib()
    g = np.random.default_rng(8)
    m, n = 9, 14
    A = g.integers(0, 10, size=(m, n)).astype(float); b = np.floor(0.5 * A.sum(axis=1)); c = g.integers(1, 21, size=n).astype(float)
    T, basis = synth.primal_tableau_from(c, A, b)
    R0, C0 = T.shape
    root = gpu.DeviceTableau.from_host(T, basis)
    depths = [0, 1, 3, 5, 2]
    cuts = [[(int(g.integers(0, n)), float(g.choice([1.0, -1.0])), float(g.choice([0.0, -0.0])), float(g.integers(0, 3))) for _ in range(d)] for d in depths]
    cap = 8
    one = [gpu.DeviceTableau(R0 + cap, C0 + cap) for _ in depths]
    many = [gpu.DeviceTableau(R0 + cap, C0 + cap) for _ in depths]
    try:
        for t, cs in zip(one, cuts):
            var = np.array([x[0] for x in cs], np.int32); coef = np.array([x[1] for x in cs]); zero = np.array([x[2] for x in cs]); rhs = np.array([x[3] for x in cs])
            gpu._lib.check(lib.lpx_tableau_build_node(t._h, root._h, len(cs), var.ctypes.data_as(gpu._lib.ip), coef.ctypes.data_as(gpu._lib.dp),
                                                      zero.ctypes.data_as(gpu._lib.dp), rhs.ctypes.data_as(gpu._lib.dp)))
        flat = [x for cs in cuts for x in cs]
        off = np.cumsum([0] + depths).astype(np.int32)
        var = np.array([x[0] for x in flat], np.int32); coef = np.array([x[1] for x in flat]); zero = np.array([x[2] for x in flat]); rhs = np.array([x[3] for x in flat])
        arr = (C.c_void_p * len(many))(*[t._h for t in many])
        gpu._lib.check(lib.lpx_tableau_build_nodes(arr, root._h, len(many), off.ctypes.data_as(gpu._lib.ip), var.ctypes.data_as(gpu._lib.ip),
                                                   coef.ctypes.data_as(gpu._lib.dp), zero.ctypes.data_as(gpu._lib.dp), rhs.ctypes.data_as(gpu._lib.dp)))
        for a, bq, d in zip(one, many, depths):
            ra, ca, rb_, cb_ = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            gpu._lib.check(lib.lpx_tableau_shape(a._h, C.byref(ra), C.byref(ca), None)); gpu._lib.check(lib.lpx_tableau_shape(bq._h, C.byref(rb_), C.byref(cb_), None))
            assert (ra.value, ca.value) == (rb_.value, cb_.value) == (R0 + d, C0 + d)
            a.R, a.C, bq.R, bq.C = ra.value, ca.value, rb_.value, cb_.value
            Ta, ba = a.download(); Tb, bb = bq.download()
            assert np.array_equal(Ta.view(np.uint64), Tb.view(np.uint64)) and ba.tolist() == bb.tolist()
    finally:
        root.close()
        for t in one + many:
            t.close()


def test_rolling_batches_suspend_and_continue_bitwise(gpu, oracle):
    """lpx_multi_run_some: a run that returns while some tableaux are still going (LPX_RUNNING = 4), handed in again beside
    fresh tableaux, ends in the same tableaux -- bit for bit -- as the oracle's solve of each LP; pivot counts are cumulative."""
    import ctypes as C
    lib = gpu._lib.lib()
    g = np.random.default_rng(21)
    lps = []
    for k in range(14):
        m, n = int(g.integers(20, 90)), int(g.integers(30, 120))
        c, A, b = g.uniform(0.5, 1.5, n), g.random((m, n)), 0.5 * n * g.uniform(0.9, 1.1, m)
        lps.append(synth.primal_tableau_from(c, A, b))
    refs = []
    for T, basis in lps:
        Tr, br = T.copy(), basis.copy()
        st, tr = oracle.primal_tableau(Tr, br)
        refs.append((st, len(tr), Tr, br))
    assert len({r[1] for r in refs}) > 4                             # different lengths: some finish early, some late
    ts = [gpu.DeviceTableau.from_host(T, basis) for T, basis in lps]
    po = gpu._lib.default_opts(False, resident=-1, batch=8); do = gpu._lib.default_opts(True, resident=-1, batch=8)
    try:
        waiting, inflight, done, calls, suspended_seen = list(range(len(ts))), [], {}, 0, 0
        while waiting or inflight:
            while waiting and len(inflight) < 6:
                inflight.append(waiting.pop(0))
            k = len(inflight)
            hs = (C.c_void_p * k)(*[ts[i]._h for i in inflight]); dl = (C.c_int * k)(*([0] * k))
            st = (C.c_int * k)(); ss = (gpu._lib.Stats * k)()
            gpu._lib.check(lib.lpx_multi_run_some(hs, dl, k, C.byref(po), C.byref(do), st, ss, 3 if waiting else 0))
            calls += 1
            keep = []
            for j, i in enumerate(inflight):
                if st[j] == 4:
                    keep.append(i); suspended_seen += 1
                else:
                    done[i] = (st[j], ss[j].pivots)
            inflight = keep
            assert calls < 200
        assert suspended_seen > 0                                       # the suspension path did run
        for i, t in enumerate(ts):
            Tg, bg = t.download()
            st_ref, piv_ref, Tr, br = refs[i]
            assert done[i] == (st_ref, piv_ref), i
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist(), i
    finally:
        for t in ts:
            t.close()


def test_rolling_batches_dual_phase_machine_survives_suspension(gpu, oracle):
    """The same through the dual path (ForceDualFeasibility -> dual loop -> repaired clean-up, Models/DualSimplex.cs:24,36-113):
    a run suspended in any phase continues in that phase with its counters; tableaux, bases, statuses and the
    ForceDualFeasibility pivot counts equal the oracle's."""
    import ctypes as C
    lib = gpu._lib.lib()
    cases = [(20, 30, 2, 5), (40, 64, 3, 10), (64, 100, 4, 7), (100, 160, 5, 30), (30, 50, 6, 8), (80, 90, 7, 20), (50, 120, 8, 12), (90, 140, 9, 25)]
    lps, refs = [], []
    for (m, n, seed, n_ge) in cases:
        c, A, b = synth.dense_lp(m, n, seed=seed)
        T, basis = synth.primal_tableau_from(c, A, b)
        g = np.random.Generator(np.random.PCG64(seed + 99))
        for i in g.choice(m, size=n_ge, replace=False):
            T[i, :n] *= -1.0
            T[i, -1] = -0.02 * T[i, -1]
        lps.append((T, basis))
        Tr, br = T.copy(), basis.copy()
        st, tr, nfdf = oracle.dual_tableau(Tr, br, fdf_guard=10000, cleanup=1)
        refs.append((st, len(tr), nfdf, Tr, br))
    ts = [gpu.DeviceTableau.from_host(T, basis) for T, basis in lps]
    po = gpu._lib.default_opts(False, resident=-1, batch=4)
    do = gpu._lib.default_opts(True, resident=-1, batch=4, fdf_guard=10000, cleanup=1)
    try:
        waiting, inflight, done, calls, suspended_seen = list(range(len(ts))), [], {}, 0, 0
        while waiting or inflight:
            while waiting and len(inflight) < 4:
                inflight.append(waiting.pop(0))
            k = len(inflight)
            hs = (C.c_void_p * k)(*[ts[i]._h for i in inflight]); dl = (C.c_int * k)(*([1] * k))
            st = (C.c_int * k)(); ss = (gpu._lib.Stats * k)()
            gpu._lib.check(lib.lpx_multi_run_some(hs, dl, k, C.byref(po), C.byref(do), st, ss, 2 if waiting else 0))
            calls += 1
            keep = []
            for j, i in enumerate(inflight):
                if st[j] == 4:
                    keep.append(i); suspended_seen += 1
                else:
                    done[i] = (st[j], ss[j].pivots, ss[j].fdf_pivots)
            inflight = keep
            assert calls < 2000
        assert suspended_seen > 0
        for i, t in enumerate(ts):
            Tg, bg = t.download()
            st_ref, piv_ref, nfdf_ref, Tr, br = refs[i]
            assert done[i] == (st_ref, piv_ref, nfdf_ref), (i, done[i], (st_ref, piv_ref, nfdf_ref))
            assert np.array_equal(Tg.view(np.uint64), Tr.view(np.uint64)) and bg.tolist() == br.tolist(), i
    finally:
        for t in ts:
            t.close()


def test_warm_search_is_the_same_search_however_the_batches_roll(gpu):
    """The rolling batches only decide WHEN a warm-started child's run continues: node log, node objective values (bitwise),
    LP count and pivot total of the warm search on a 0/1 program whose node tableaux are 1.2 MB are the same whether, on the
    streaming kernels, a run is suspended as soon as half of the batch is left, only when one node is left, or polled every 4
    pivots -- and the same again on the resident group kernel, which is where such nodes run by default since r03."""
    import subprocess, sys, os, textwrap, json
    code = textwrap.dedent('''
        import json, numpy as np
        import linear_programming_solver_lpr381_amd as L
        from linear_programming_solver_lpr381_amd import synth
        c, A, rel, b = synth.binary_ip(200, 100, seed=5)
        p = L.LPProblem.from_arrays(0, c, A, rel, b)
        r = L.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=16, max_nodes=400).Solve(p)
        print(json.dumps([r.LpSolves, int(r.Stats["pivots"]), np.asarray(r.NodeLog).tolist(), np.asarray(r.NodeZ, np.float64).view(np.uint64).tolist()]))
    ''')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # the streaming rolling batches three ways (LPX_WARM_RESIDENT=0) and the default: the same nodes on the resident group kernel
    for env in ({"LPX_WARM_RESIDENT": "0"}, {"LPX_WARM_RESIDENT": "0", "LPX_ROLL_DIV": "1000000"},
                {"LPX_WARM_RESIDENT": "0", "LPX_ROLL_BATCH": "4", "LPX_ROLL_DIV": "3"}, {}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=root, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0][0] > 300 and outs[0][1] > 1000
    assert outs[0] == outs[1] == outs[2] == outs[3]
