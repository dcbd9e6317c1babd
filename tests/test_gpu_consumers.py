"""GPU parity for the consumers of SimplexResult.Tableau / Basis / VarNames:
CuttingPlane, CuttingPlaneRevised (Models/CuttingPlane*.cs) and SensitivityAnalysis
(Models/SensitivityAnalysis.cs).  Every LP inside them runs on the HIP pivot loops; expected texts are derived
by hand from the cited lines (tests/test_oracle_kats.py holds the derivations), numbers come from the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LE_SIGN = " ≤ "


def _oracle_problem(oracle, p):
    A = np.array([c.A for c in p.Constraints], dtype=float)
    return oracle.Problem(int(p.ObjectiveSense), p.C, A, [int(c.Relation) for c in p.Constraints],
                          [c.B for c in p.Constraints])


def _ip(gpu, g, n, m):
    A = g.integers(0, 7, size=(m, n)).astype(float)
    b = np.floor(0.6 * A.sum(axis=1)) + 0.5 * g.integers(0, 2, size=m)
    c = g.integers(1, 9, size=n).astype(float)
    return gpu.LPProblem.from_arrays(0, c, A, np.zeros(m, int), b)


# ---- CuttingPlane ---------------------------------------------------------------------------------------
def test_kat10_cutting_plane_text_and_defect(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    p = P(gpu.Sense.Max, [1.0], [K([2.0], gpu.Rel.LE, 3.0)])
    r = gpu.CuttingPlane().Solve(p)
    assert r.Status == gpu._lib.CUT_INCOMPLETE and r.Summary == "Status: INCOMPLETE" and r.LpSolves == 50
    assert r.Cuts.tolist() == [[0.0, 0.5]] * 50 and r.Solution is None
    assert r.Report.startswith("=== Gomory Cutting Plane Algorithm ===\nObjective: Maximize 1.000x1\nSubject to:\n"
                               "2.000x1 LE 3.000\nx_j >= 0, integer\n\n--- Iteration 1 ---\nObjective: max +1x1\n")
    assert "Current solution: x* = [1.500], z* = 1.500\nAdded Gomory cut:  <= 0.500\n\n--- Iteration 2 ---\n" in r.Report
    assert r.Report.endswith("Added Gomory cut:  <= 0.500\nIteration limit reached. Stopping.\n")
    assert len(p.Constraints) == 1          # the caller's model is cloned, never mutated (:15)


def test_kat11_cutting_plane_integer_summary(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    p = P(gpu.Sense.Max, [1.0, 1.0], [K([1.0, 0.0], gpu.Rel.LE, 2.0), K([0.0, 1.0], gpu.Rel.LE, 3.0)])
    r = gpu.CuttingPlane().Solve(p)
    assert r.Status == gpu._lib.CUT_INTEGER and r.Cuts is None
    assert r.Summary == "Status: OPTIMAL INTEGER\nz* = 5.00\nx* = [2.00, 3.00]"
    assert r.OptimalValue == 5.0 and r.Solution.tolist() == [2.0, 3.0]
    assert r.Tableau.shape == (3, 5) and r.Basis.tolist() == [0, 1] and r.VarNames == ["x1", "x2", "c1", "c2"]
    assert r.Report.endswith("Current solution: x* = [2.000, 3.000], z* = 5.000\n"
                             "All variables integer. Optimal integer solution found.\n")
    assert "1.000x1 LE 2.000\n1.000x2 LE 3.000\n" in r.Report       # zero coefficients are dropped (:28)


def test_cutting_plane_error_paths(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    r = gpu.CuttingPlane().Solve(P(gpu.Sense.Max, [1.0], [K([1.0], gpu.Rel.GE, 1.0)]))
    assert r.Status == gpu._lib.CUT_ERROR and r.Summary.startswith("Error: Constraint contains '>=' sign")
    assert "Error in PrimalSimplex: Constraint contains '>=' sign" in r.Report and r.Solution is None
    with pytest.raises(gpu.SolverException, match="supports only <= constraints"):     # :23, not caught
        gpu.CuttingPlaneRevised().Solve(P(gpu.Sense.Max, [1.0], [K([1.0], gpu.Rel.GE, 1.0)]))
    # unbounded LP: CuttingPlane goes on with the tableau it was handed; the revised one stops
    r = gpu.CuttingPlaneRevised().Solve(P(gpu.Sense.Max, [1.0, 1.0], [K([1.0, -1.0], gpu.Rel.LE, 1.0)]))
    assert r.Status == gpu._lib.CUT_NOT_OPTIMAL and r.Summary == "Terminated: LP not OPTIMAL; cutting-plane stopped."


def test_cutting_plane_random_matches_oracle_bitwise(gpu, oracle):
    g = np.random.default_rng(77)
    seen = set()
    for trial in range(12):
        p = _ip(gpu, g, int(g.integers(2, 6)), int(g.integers(2, 5)))
        ref = oracle.cutting_plane(_oracle_problem(oracle, p))
        r = gpu.CuttingPlane().Solve(p)
        want = {oracle.CUT_INTEGER: gpu._lib.CUT_INTEGER, oracle.CUT_INCOMPLETE: gpu._lib.CUT_INCOMPLETE,
                oracle.CUT_ERROR: gpu._lib.CUT_ERROR, oracle.CUT_NONBASIC: gpu._lib.CUT_ERROR}[ref.status]
        assert r.Status == want, trial
        cuts = r.Cuts if r.Cuts is not None else np.zeros((0, p.NumVars + 1))
        assert np.array_equal(cuts.view(np.uint64), ref.cuts.view(np.uint64)), trial      # bit for bit
        assert r.LpSolves == ref.lp_solves and r.Stats["pivots"] == ref.total_pivots
        if ref.status == oracle.CUT_INTEGER:
            assert r.OptimalValue == ref.z and r.Solution.tolist() == ref.x.tolist()
        seen.add(ref.status)
    assert len(seen) >= 2, seen


def test_kat12_cutting_plane_revised_text(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    r = gpu.CuttingPlaneRevised().Solve(P(gpu.Sense.Max, [1.0], [K([2.0], gpu.Rel.LE, 3.0)]))
    assert r.Status == gpu._lib.CUT_INTEGER and r.Cuts.tolist() == [[1.0, 1.0]] and r.LpSolves == 2
    assert r.Summary == "Status: OPTIMAL INTEGER\nx* = [1]\nz* = 1\n"
    assert r.Report == ("--- Cutting-Plane Iteration 1 ---\n\nStatus: OPTIMAL\n  x1 = 1.5\n  z* = 1.5\n\n"
                        "Added cut: x1 ≤ 1 (current x1 = 1.5)\n"
                        "--- Cutting-Plane Iteration 2 ---\n\nStatus: OPTIMAL\n  x1 = 1\n  z* = 1\n\n"
                        "All decision variables are integer. Optimal integer solution found.\n")


def test_cutting_plane_revised_random_matches_oracle(gpu, oracle):
    g = np.random.default_rng(78)
    done = 0
    for trial in range(10):
        p = _ip(gpu, g, int(g.integers(2, 5)), int(g.integers(2, 4)))
        ref = oracle.cutting_plane(_oracle_problem(oracle, p), revised=True)
        r = gpu.CuttingPlaneRevised().Solve(p)
        want = {oracle.CUT_INTEGER: gpu._lib.CUT_INTEGER, oracle.CUT_INCOMPLETE: gpu._lib.CUT_INCOMPLETE,
                oracle.CUT_NOT_OPTIMAL: gpu._lib.CUT_NOT_OPTIMAL}[ref.status]
        cuts = r.Cuts if r.Cuts is not None else np.zeros((0, p.NumVars + 1))
        # the 3-decimal round trip of x* (:90-110) absorbs the 1e-9 differences of the revised path
        assert r.Status == want and cuts.tolist() == ref.cuts.tolist(), trial
        assert r.LpSolves == ref.lp_solves
        if ref.status != oracle.CUT_NOT_OPTIMAL:
            assert r.Extra.tolist() == ref.x.tolist()
        done += ref.status == oracle.CUT_INTEGER
    assert done >= 3


# ---- SensitivityAnalysis --------------------------------------------------------------------------------
def test_kat13_sensitivity_reports_on_kat1(gpu):
    p = gpu.ParseFromText("Max: 3x1 + 5x2\n1x1 + 0x2 <= 4\n0x1 + 2x2 <= 12\n3x1 + 2x2 <= 18\n")
    res = gpu.PrimalSimplex().Solve(p)
    sa = gpu.SensitivityAnalysis(p, res)
    assert sa.GetRangeReport("x1") == "x1 (Basic): -∞" + LE_SIGN + "c" + LE_SIGN + "2.778"
    assert sa.GetRangeReport("x2") == "x2 (Basic): 6.000" + LE_SIGN + "c" + LE_SIGN + "6.000"
    assert sa.GetRangeReport("Constraint 1") == "Constraint 1: -∞" + LE_SIGN + "B" + LE_SIGN + "∞"
    assert sa.GetShadowPricesReport() == "Shadow Prices:\n  Constraint 1: -1.000\n  Constraint 2: -0.333\n  Constraint 3: 0.333\n"
    for target in ("c1", "c2"):                                   # problem.C[col], col >= NumVars (:236, :256)
        with pytest.raises(gpu.SolverException, match="outside the bounds of the array"):
            sa.GetRangeReport(target)
    with pytest.raises(gpu.SolverException, match="Variable 'y1' not found in VarNames"):
        sa.GetRangeReport("y1")
    with pytest.raises(gpu.SolverException, match="Invalid constraint index in 'Constraint 4'. Expected 1 to 3"):
        sa.GetRangeReport("Constraint 4")
    with pytest.raises(gpu.SolverException, match="Target cannot be empty"):
        sa.GetRangeReport("  ")


def test_kat14_sensitivity_nonbasic_and_apply_change(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    p = P(gpu.Sense.Max, [3.0, 1.0], [K([1.0, 1.0], gpu.Rel.LE, 4.0)])
    res = gpu.PrimalSimplex().Solve(p)
    sa = gpu.SensitivityAnalysis(p, res)
    assert sa.GetRangeReport("x2") == "x2 (Non-Basic): -∞" + LE_SIGN + "c" + LE_SIGN + "2.000"
    assert sa.GetRangeReport("Constraint 1") == "Constraint 1: -∞" + LE_SIGN + "B" + LE_SIGN + "6.000"
    assert sa.GetRangeReport("x1") == "x1 (Basic): -∞" + LE_SIGN + "c" + LE_SIGN + "2.500"
    assert sa.ApplyChange("x2", 7.25) == "Non-basic variable x2 objective coefficient updated to 7.250"
    assert sa.ApplyChange("x1", -1) == "Basic variable x1 objective coefficient updated to -1.000"
    assert sa.ApplyChange("Constraint 1", 2.0625) == "Constraint 1 B-value updated to 2.063"     # half away from zero
    assert list(p.C) == [-1.0, 7.25] and p.Constraints[0].B == 2.0625                # the caller's model IS mutated


def test_sensitivity_constructor_checks(gpu):
    P, K = gpu.LPProblem, gpu.Constraint
    p = P(gpu.Sense.Max, [3.0, 1.0], [K([1.0, 1.0], gpu.Rel.LE, 4.0)])
    rev = gpu.RevisedPrimalSimplex().Solve(p)                      # text only (:294)
    with pytest.raises(gpu.SolverException, match="SimplexResult.Tableau cannot be null"):
        gpu.SensitivityAnalysis(p, rev)
    res = gpu.PrimalSimplex().Solve(p)
    p2 = P(gpu.Sense.Max, [3.0, 1.0], [K([1.0, 1.0], gpu.Rel.LE, 4.0), K([1.0, 0.0], gpu.Rel.LE, 1.0)])
    with pytest.raises(gpu.SolverException, match="Tableau dimensions invalid. Expected 3 rows, 5 columns, got 2 rows, 4 columns."):
        gpu.SensitivityAnalysis(p2, res)
    with pytest.raises(gpu.SolverException, match="Parameter 'result'"):
        gpu.SensitivityAnalysis(p, None)


def test_sensitivity_ranges_random_match_oracle_bitwise(gpu, oracle):
    g = np.random.default_rng(79)
    kinds = set()
    for trial in range(8):
        n, m = int(g.integers(2, 7)), int(g.integers(2, 6))
        A = g.uniform(0, 1, size=(m, n)); b = g.uniform(1, 3, size=m); c = g.uniform(0.2, 2, size=n)
        p = gpu.LPProblem.from_arrays(0, c, A, np.zeros(m, int), b)
        res = gpu.PrimalSimplex().Solve(p)
        op = _oracle_problem(oracle, p)
        ores = oracle.primal_solve(op)
        assert np.array_equal(res.Tableau, ores.T)
        sa = gpu.SensitivityAnalysis(p, res)
        for j in range(n):
            rc, mn, mx, which = oracle.sens_range(op, ores.T, ores.basis, 1, j)
            assert rc == 0 and sa.GetRange(f"x{j + 1}") == (mn, mx), (trial, j)
            kinds.add(which)
            tag = "Basic" if which == 1 else "Non-Basic"
            assert sa.GetRangeReport(f"x{j + 1}").startswith(f"x{j + 1} ({tag}): ")
        for i in range(m):
            rc, mn, mx, _ = oracle.sens_range(op, ores.T, ores.basis, 0, i)
            assert sa.GetRange(f"Constraint {i + 1}") == (mn, mx), (trial, i)
        sh = oracle.sens_shadow_prices(op, ores.T)
        lines = sa.GetShadowPricesReport().splitlines()
        assert lines[0] == "Shadow Prices:" and len(lines) == m + 1
        for i in range(m):
            assert abs(float(lines[i + 1].split(": ")[1]) - sh[i]) <= 0.0005 + 1e-12
    assert kinds == {1, 2}


def test_solve_using_duality(gpu, oracle):
    p = gpu.ParseFromText("Max: 3x1 + 5x2\n1x1 + 0x2 <= 4\n0x1 + 2x2 <= 12\n3x1 + 2x2 <= 18\n")
    res = gpu.PrimalSimplex().Solve(p)
    # faithful DualSimplex returns text only (defect D2): w* = 0, y* = [] (:205-207)
    d = gpu.SensitivityAnalysis(p, res).SolveUsingDuality()
    assert d.Report.startswith("=== Duality Algorithm Solution ===\nDual Problem: Minimize 4.000y1 + 12.000y2 + 18.000y3\n"
                               "Subject to:\n1.000y1 + 0.000y2 + 3.000y3 GE 3.000\n0.000y1 + 2.000y2 + 2.000y3 GE 5.000\n"
                               "\n=== Duality Algorithm Iterations ===\n")
    assert "Debug: Raw Solution = []\nDebug: VarNames = []\n\n=== Final Result ===\n" in d.Report
    assert d.Report.endswith("w* = 0.000\ny* = []\n") and d.Tableau is None and d.OptimalValue == 0.0
    # the model handed to DualSimplex: C = b, rows = columns of A, all >=, sense left at Max (:136-160)
    dual = oracle.Problem(oracle.MAX, [4, 12, 18.0], [[1, 0, 3], [0, 2, 2.0]], [oracle.GE, oracle.GE], [3, 5.0])
    ref = oracle.dual_solve(dual, 0)
    want = "OPTIMAL" if ref.status == 0 else "INFEASIBLE"
    assert f"\n=== Final Result ===\nStatus: {want}\n" in d.Report
    # repaired DualSimplex: numbers come back and must equal the oracle's
    d = gpu.SensitivityAnalysis(p, res, dual_flags=7).SolveUsingDuality()
    ref = oracle.dual_solve(dual, 7)
    assert d.Status == ref.status
    if ref.has_solution:
        assert d.OptimalValue == ref.z and d.Solution.tolist() == ref.x[:2].tolist()
        assert np.array_equal(d.Tableau, ref.T) and d.VarNames[:3] == ["x1", "x2", "x3"]
