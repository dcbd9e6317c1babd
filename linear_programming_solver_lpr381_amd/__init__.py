"""lpx -- MI355X (gfx950) simplex / branch-and-bound engine behind the solver entry points of
Jellyman750/Linear_Programming_Solver_LPR381 (Models/PrimalSimplex.cs, RevisedPrimalSimplex.cs,
Branch&Bound.cs, BranchAndBoundKnapsack.cs).  The compute path is liblpx.so (hand-written HIP
behind the C ABI of include/lpx.h); this package is the host-side binding and has no CPU fallback.
"""
from . import _lib
from . import comm
from ._lib import LpxError, default_opts
from .tableau import DeviceTableau, primal_tableau, dual_tableau, multi_run
from .revised import DeviceRevised, invert
from .solver import (BranchAndBound, BranchAndBoundKnapsack, BranchAndBoundRevised, Constraint, CuttingPlane,
                     CuttingPlaneRevised, DeviceKnapsack, DualSimplex, LPProblem, SensitivityAnalysis,
                     LPSolver, ParseFromText, PrimalSimplex, Rel, RevisedPrimalSimplex, Sense, SimplexResult,
                     SolverException)

__all__ = ["_lib", "comm", "LpxError", "default_opts", "DeviceTableau", "primal_tableau", "dual_tableau", "multi_run", "DeviceRevised", "invert", "LPSolver", "LPProblem", "Constraint", "Sense", "Rel",
           "SimplexResult", "SolverException", "PrimalSimplex", "RevisedPrimalSimplex", "DualSimplex",
           "BranchAndBound", "BranchAndBoundKnapsack", "BranchAndBoundRevised", "ParseFromText", "DeviceKnapsack",
           "CuttingPlane", "CuttingPlaneRevised", "SensitivityAnalysis"]
