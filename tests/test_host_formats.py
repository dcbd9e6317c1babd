"""CPU tests of the host mirror's text side: the reference's input grammar
(Models/LPParser.cs) and number / tableau formats (Models/PrimalSimplex.cs:272-304) -- no GPU needed.
Expected strings are what .NET (Core 3.0+) prints for ToString("0.###") / "F3"."""
import numpy as np
import pytest


def test_format_number_matches_dotnet_custom_format(lpx):
    f = lpx.solver.format_number
    assert f(0.0) == "0" and f(2.0) == "2" and f(2.5) == "2.5" and f(36.0) == "36"
    assert f(1234.5678) == "1234.568" and f(0.001) == "0.001" and f(0.0004) == "0"
    assert f(-0.0004) == "-0"                       # .NET Core 3.0+ keeps the sign of a value that rounds to zero
    assert f(-3.14159) == "-3.142" and f(1e6) == "1000000"
    # exact decimal midpoints (dyadic rationals) round AWAY from zero, not to even
    assert f(0.0625) == "0.063" and f(0.1875) == "0.188" and f(-0.0625) == "-0.063" and f(2.0625) == "2.063"
    assert f(0.5) == "0.5" and f(0.125) == "0.125"
    # not midpoints in binary: plain nearest
    assert f(0.0005) == "0.001" or f(0.0005) == "0"   # 0.0005 is not exactly representable; either side of it is accepted
    assert f(float("nan")) == "NaN" and f(float("inf")) == "∞" and f(float("-inf")) == "-∞"


def test_parser_matches_oracle_on_reference_grammar(lpx, oracle):
    texts = [
        "Max: 3x1 + 5x2\n1x1 + 0x2 <= 4\n0x1 + 2x2 <= 12\n3x1 + 2x2 <= 18\n",
        "  max :  x1 - x2 + 2.5x3 \r\n\r\n -x1 + x2 - 0.5x3 >= -4\nx1+x2+x3=3\n",
        "MIN: -x1 - .5x2\n2x1 + 3.x2 <= 1e1\n",
        "Max: 3x1 + 2x3\n1x9 <= 4\n",
    ]
    for t in texts:
        ref, ragged = oracle.parse_text(t)
        p = lpx.ParseFromText(t)
        assert int(p.ObjectiveSense) == ref.sense and list(p.C) == ref.c.tolist()
        assert [int(c.Relation) for c in p.Constraints] == ref.rel.tolist()
        assert [c.B for c in p.Constraints] == ref.b.tolist()
        assert p.ragged == ragged
        for c, row in zip(p.Constraints, ref.A):
            assert list(c.A)[: len(ref.c)] == row.tolist()[: len(c.A)] or ragged


@pytest.mark.parametrize("bad,msg", [
    ("Max: 3x1\n", "Input must contain an objective and at least one constraint."),
    ("Maximize 3x1\nx1<=1\n", "Objective format incorrect. Example: Max: 3x1 + 5x2"),
    ("Max: 3y1\nx1<=1\n", "Cannot parse coefficient: 3y1"),
    ("Max: 3x1\nx1 < 1\n", "Constraint format incorrect: x1 < 1"),
    ("Max: 3x1\nx1 <= abc\n", "Invalid RHS number: abc"),
    ("Max: .x1\nx1 <= 1\n", "Cannot parse coefficient: .x1"),
])
def test_parser_errors_carry_reference_messages(lpx, bad, msg):
    with pytest.raises(lpx.SolverException) as e:
        lpx.ParseFromText(bad)
    assert str(e.value) == msg and e.value.code == lpx._lib.E_PARSE


def test_unknown_and_blank_algorithm_keys_fail_before_touching_the_gpu(lpx):
    p = lpx.ParseFromText("Max: x1\nx1 <= 1\n")
    with pytest.raises(lpx.SolverException, match="Algorithm not supported: 'simplex\\+\\+'") as e:
        lpx.LPSolver().Solve(p, "simplex++")
    assert e.value.code == lpx._lib.E_UNKNOWN_ALGO
    with pytest.raises(lpx.SolverException, match="No algorithm selected."):
        lpx.LPSolver().Solve(p, " \t ")


def test_model_preconditions_fail_before_touching_the_gpu(lpx):
    P, C_, S, R = lpx.LPProblem, lpx.Constraint, lpx.Sense, lpx.Rel
    with pytest.raises(lpx.SolverException) as e:
        lpx.PrimalSimplex().Solve(P(S.Max, [1, 1], [C_([1, 1], R.GE, 1)]))
    assert e.value.code == lpx._lib.E_GE_PRESENT
    with pytest.raises(lpx.SolverException) as e:
        lpx.PrimalSimplex().Solve(P(S.Max, [1, 1], [C_([1, 1], R.LE, -1)]))
    assert e.value.code == lpx._lib.E_NEG_RHS
    with pytest.raises(lpx.SolverException) as e:
        lpx.RevisedPrimalSimplex().Solve(P(S.Max, [1], [C_([1], R.GE, 1)]))
    assert e.value.code == lpx._lib.E_REVISED_PRECOND
    with pytest.raises(lpx.SolverException) as e:
        lpx.BranchAndBoundKnapsack().Solve(P(S.Max, [1, 2], [C_([1, 1], R.GE, 1)]))
    assert e.value.code == lpx._lib.E_KNAP_SHAPE
    with pytest.raises(lpx.SolverException, match="Index was outside the bounds of the array."):
        lpx.PrimalSimplex().Solve(P(S.Max, [1, 1], [C_([1], R.LE, 1)]))     # ragged row, PrimalSimplex.cs:190
