"""Worker of tests/test_distributed_cpu.py: one rank of a world_size-2 gloo job on CPU.

The sharded host logic under test is the product's (csrc/host/bnb.cpp, knapsack.cpp); the device
loops are stood in for by the CPU oracle through the documented test seams of include/lpx.h, and the
per-level exchange is torch.distributed.all_reduce(MAX) over gloo (RCCL on the GPU box).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import linear_programming_solver_lpr381_amd as L
from oracle import oracle as O


def node_lp(_u, T, R, Cc, basis, dual, repaired, max_iter, nvars, x, z, pivots):
    Tn = np.ctypeslib.as_array(T, shape=(R, Cc))
    bn = np.ctypeslib.as_array(basis, shape=(R - 1,))
    if dual:
        st, tr, _ = O.dual_tableau(Tn, bn, fdf_guard=max_iter if repaired else 100, max_iter=max_iter,
                                   cleanup=1 if repaired else 0)
    else:
        st, tr = O.primal_tableau(Tn, bn, max_iter=max_iter)
    xs = np.ctypeslib.as_array(x, shape=(nvars,))
    xs[:] = 0.0
    for i in range(R - 1):
        if bn[i] < nvars:
            xs[bn[i]] = Tn[i, Cc - 1]
    z[0] = Tn[R - 1, Cc - 1]
    pivots[0] = len(tr)
    return int(st)


def make_knap_relax(profit, weight, cap):
    order = O.knapsack_order(profit, weight)
    n = len(profit)

    def relax(_u, count, off, fidx, fval, p, w, fr, fx):
        for k in range(count):
            a = -np.ones(n, np.int32)
            for e in range(off[k], off[k + 1]):
                a[fidx[e]] = fval[e]
            rp, rw, rf, rx = O.knapsack_relax(profit, weight, cap, order, a, want_vector=True)
            p[k], w[k], fr[k] = rp, rw, rf
            fx[k] = rx[order[rf]] if rf >= 0 else 0.0
        return 0
    return relax


def main():
    import torch
    import torch.distributed as dist
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = {"n": 0}

    def allreduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        calls["n"] += 1
        return t.numpy()

    res = {"rank": rank}
    # --- branch and bound, repaired mode, sharded level search ---
    g = np.random.default_rng(31)
    n, m = 14, 6
    A = g.integers(0, 10, size=(m, n)).astype(float)
    b = np.floor(0.5 * A.sum(axis=1))
    c = g.integers(1, 21, size=n).astype(float)
    Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
    p = L.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
    r = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, rank=rank, world=world,
                         allreduce_max=allreduce_max, test_node_lp=node_lp).Solve(p)
    res["bnb"] = {"z": r.OptimalValue, "x": r.Solution.tolist(), "lp_solves": r.LpSolves, "nodes": r.Nodes,
                  "allreduces": calls["n"], "aux": list(r.Aux)}
    if rank == 0:
        one = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, test_node_lp=node_lp).Solve(p)
        res["bnb_single"] = {"z": one.OptimalValue, "lp_solves": one.LpSolves}
    ref = O.bnb_solve(O.Problem(O.MAX, c, Af, np.zeros(m + n, np.int32), bf), 1)
    res["bnb_ref"] = {"z": ref.best_z, "nodes": ref.nodes_visited}
    # --- tiny global node budgets: the budget trips while the warm-up is still replicated, or right after the hand-out;
    #     both ranks must leave together (same number of collectives), whatever the budget
    res["budget"] = []
    for cap in (1, 2, 3, 4, 7, 12):
        calls["n"] = 0
        rb = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, rank=rank, world=world, max_nodes=cap,
                              allreduce_max=allreduce_max, test_node_lp=node_lp).Solve(p)
        res["budget"].append({"cap": cap, "nodes": rb.Nodes, "allreduces": calls["n"], "z": rb.OptimalValue})
    # --- equal-z ties: two interchangeable variables (same column, same cost) give pairs of optimal solutions in
    #     different subtrees; the published x must be the one the single-process search keeps (DFS-order key), on both ranks
    A2 = np.hstack([A, A[:, :1]]); c2 = np.concatenate([c, c[:1]]); n2 = n + 1
    Af2 = np.vstack([A2, np.eye(n2)]); bf2 = np.concatenate([b, np.ones(n2)])
    p2 = L.LPProblem.from_arrays(0, c2, Af2, np.zeros(m + n2, int), bf2)
    calls["n"] = 0
    rt = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, rank=rank, world=world,
                          allreduce_max=allreduce_max, test_node_lp=node_lp).Solve(p2)
    one2 = L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, test_node_lp=node_lp).Solve(p2)
    ref2 = O.bnb_solve(O.Problem(O.MAX, c2, Af2, np.zeros(m + n2, np.int32), bf2), 1)
    res["ties"] = {"z": rt.OptimalValue, "x": rt.Solution.tolist(), "single_x": one2.Solution.tolist(), "single_z": one2.OptimalValue,
                   "dfs_x": ref2.best_x.tolist(), "dfs_z": ref2.best_z, "aux": list(rt.Aux), "lp_solves": rt.LpSolves}
    # --- rebalancing: depth-first-K pools of different fortunes; descriptors must move to the rank that runs dry
    calls["n"] = 0
    rr = L.BranchAndBound(bnb_mode=1, bnb_search=1, bnb_dive=1, concurrent_nodes=1, rank=rank, world=world,
                          allreduce_max=allreduce_max, test_node_lp=node_lp).Solve(p)
    res["rebalance"] = {"z": rr.OptimalValue, "x": rr.Solution.tolist(), "aux": list(rr.Aux), "lp_solves": rr.LpSolves,
                        "allreduces": calls["n"]}
    # --- knapsack, sharded rounds ---
    calls["n"] = 0
    kn = 60
    w = g.integers(1, 60, size=kn).astype(float)
    pr = w + g.integers(0, 12, size=kn)
    cap = float(np.floor(0.5 * w.sum()))
    kp = L.LPProblem(L.Sense.Max, pr.tolist(), [L.Constraint(w.tolist(), L.Rel.LE, cap)])
    kr = L.BranchAndBoundKnapsack(rank=rank, world=world, allreduce_max=allreduce_max,
                                  test_knap_relax=make_knap_relax(pr, w, cap)).Solve(kp)
    kref = O.knapsack_solve(O.Problem(O.MAX, pr, w.reshape(1, -1), [O.LE], [cap]))
    res["knap"] = {"z": kr.OptimalValue, "x": kr.Extra.astype(int).tolist(), "popped": kr.Nodes, "allreduces": calls["n"],
                   "feasible": bool((kr.Extra * w).sum() <= cap + 1e-9), "value": float((kr.Extra * pr).sum())}
    res["knap_ref"] = {"z": kref.best_z, "popped": kref.nodes_popped}
    # --- a rank whose node group fails (LPX_ENOMEM through the test seam): BOTH ranks come back with an error, nobody waits in
    #     a collective for good -- after the hand-out (rank 1 fails at its 6th node) and inside the replicated warm-up (2nd node)
    res["peer_failure"] = []
    for fail_at in (6, 2):
        calls["n"] = 0
        kw = {"test_fail_after_nodes": fail_at} if rank == 1 else {}
        try:
            L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, rank=rank, world=world,
                             allreduce_max=allreduce_max, test_node_lp=node_lp, **kw).Solve(p)
            res["peer_failure"].append({"error": None, "allreduces": calls["n"]})
        except L.SolverException as e:
            res["peer_failure"].append({"error": str(e), "code": e.code, "allreduces": calls["n"]})
    # --- ranks whose replicated warm-up DIFFERS (here: rank 1 is handed another objective; in round 2 it was a kernel whose result
    #     depended on workgroup scheduling): the fingerprint in the first all-reduce tells every rank, nobody waits in a collective
    c_alt = c.copy(); c_alt[:3] += 7.0
    p_alt = L.LPProblem.from_arrays(0, c_alt if rank == 1 else c, Af, np.zeros(m + n, int), bf)
    calls["n"] = 0
    try:
        L.BranchAndBound(bnb_mode=1, bnb_search=1, concurrent_nodes=2, rank=rank, world=world,
                         allreduce_max=allreduce_max, test_node_lp=node_lp).Solve(p_alt)
        res["divergence"] = {"error": None, "allreduces": calls["n"]}
    except L.SolverException as e:
        res["divergence"] = {"error": str(e), "allreduces": calls["n"]}
    dist.barrier()
    dist.destroy_process_group()
    json.dump(res, open(out, "w"))


if __name__ == "__main__":
    main()
