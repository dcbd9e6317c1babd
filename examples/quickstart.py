"""The reference's own flow on the GPU engine: parse an input file in its text format (Models/LPParser.cs), solve it with the
algorithm names of LPSolver (Models/LPSolver.cs:16-76), print the result record.  Needs an MI355X: there is no CPU fallback.
(As in the reference, the revised algorithm fills only the text fields of the record: its numbers are in `Summary` / `Extra`.)

    python examples/quickstart.py [input file]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import linear_programming_solver_lpr381_amd as lpx

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "integration", "Input", "example_input.txt")
problem = lpx.ParseFromText(open(path).read())
solver = lpx.LPSolver()
for name in ("Primal Simplex", "Revised Primal Simplex", "Branch and Bound"):
    res = solver.Solve(problem, name)
    x = None if res.Solution is None else [float(v) for v in res.Solution]
    print(f"{name}: status {res.Status}, z = {res.OptimalValue}, x = {x}")
    print(res.Summary.strip().splitlines()[-1] if res.Summary else "")
