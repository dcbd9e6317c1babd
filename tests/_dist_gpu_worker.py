"""Worker of tests/test_gpu_distributed.py: one rank of a world_size-2 job whose ranks SHARE the visible GPU.  The searches,
the device loops and the node store are the product's; the per-level exchange is all_reduce(MAX) over gloo on CPU tensors
(the rehearsal backend of bench.py -- on a multi-GPU node the same callback runs over RCCL)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth


def main():
    import torch
    import torch.distributed as dist
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L._lib.check(L._lib.lib().lpx_init(0))
    calls = {"n": 0}

    def allreduce_max(vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        calls["n"] += 1
        return t.numpy()

    res = {"rank": rank}
    # Both ranks stream the headline LP (403 MB tableau, fused launch) through the SHARED GPU at the same time: workgroups of a
    # launch are then dispatched across context switches -- the situation in which a fused kernel that picked its state
    # record by a device-side comparison ended a solve after 3 pivots.
    import hashlib
    c, A, b = synth.dense_lp(4096, 8192)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    with L.DeviceTableau.from_host(T, basis) as dt:
        dist.barrier()
        status, st = dt.primal_run(max_iter=3000)
        tr = dt.trace()
        _, bg = dt.download()
    res["primal"] = {"status": int(status), "pivots": int(st["pivots"]), "launches": int(st["launches"]),
                     "trace150": np.asarray(tr[:150]).tolist(),
                     "trace_sha": hashlib.sha256(np.ascontiguousarray(tr, dtype=np.int32).view(np.uint8)).hexdigest(),
                     "basis_sha": hashlib.sha256(np.ascontiguousarray(bg, dtype=np.int32).view(np.uint8)).hexdigest()}
    del T
    sha = lambda a, dt=None: hashlib.sha256(np.ascontiguousarray(a, dtype=dt).view(np.uint8)).hexdigest()
    # ---- the other launch types under the same sharing (workgroups of one launch dispatched across context switches): each
    #      must give the bits a solo run gives.  Both ranks enter each section together (barrier) so the launches interleave.
    # (a) revised loop at config-3 size: rv_price / rv_pick / rv_upd_ftran (lazy pending update of W) / rv_select2
    c3, A3, b3 = synth.dense_lp(4096, 8192)
    with L.DeviceRevised(A3, -c3, b3) as rv:
        dist.barrier()
        status, st = rv.run(max_iter=300, batch=50)
        Bidx, Nidx, xB, z = rv.result()
        tr = rv.trace()
    res["revised"] = {"status": int(status), "pivots": int(st["pivots"]), "trace_sha": sha(tr, np.int32), "bidx_sha": sha(Bidx, np.int32),
                      "nidx_sha": sha(Nidx, np.int32), "xb_sha": sha(xB, np.float64), "z_hex": float(z).hex()}
    del A3
    # (b) dual streaming pair lpx_select / lpx_update_s (`st` / `us` records) on a 361 MB tableau
    m, n = 4500, 5500
    cd, Ad, bd = synth.dense_lp(m, n, seed=11)
    Td, basd = synth.primal_tableau_from(cd, Ad, bd)
    del Ad
    g = np.random.Generator(np.random.PCG64(11))
    for i in g.choice(m, size=12, replace=False):
        Td[i, :n] *= -1.0
        Td[i, -1] = -0.02 * Td[i, -1]
    with L.DeviceTableau.from_host(Td, basd) as dt:
        dist.barrier()
        status, st = dt.dual_run(fdf_guard=4, cleanup=1, max_iter=6)
        tr = dt.trace()
        Tg, bg = dt.download()
    res["dual"] = {"status": int(status), "fdf": int(st["fdf_pivots"]), "trace": np.asarray(tr).tolist(), "basis_sha": sha(bg, np.int32),
                   "T_sha": sha(Tg, np.float64)}
    del Td, Tg
    # (c) forced pivots on the two-launch kernels lpx_select_mb + lpx_update_mb_m at 4097 x 12289
    T0 = synth.raw_tableau(4097, 12289)
    rows, cols = synth.forced_pivot_list(4097, 12289, 8)
    with L.DeviceTableau.from_host(T0) as dt:
        dist.barrier()
        chosen, st = dt.forced_pivots(rows, cols, 0.1)
        Tg, _ = dt.download()
    res["forced"] = {"chosen": np.asarray(chosen).tolist(), "pivots": int(st["pivots"]), "T_sha": sha(Tg, np.float64)}
    del T0, Tg
    # (d) one exact K7' (inv_select + lpx_update) at n = 600
    gi = np.random.default_rng(6)
    M = gi.uniform(-1, 1, size=(600, 600))
    M[2, 0] = -M[1, 0]
    dist.barrier()
    res["invert"] = {"inv_sha": sha(L.invert(M), np.float64)}
    # a 0/1 IP small enough to be solved: cold level search with the depth-first-K pool, then the warm-started one
    cs, As, rels, bs = synth.binary_ip(24, 8, seed=11)
    ps = L.LPProblem.from_arrays(0, cs, As, rels, bs)
    for name, kw in (("cold", dict(bnb_search=1, bnb_dive=1, concurrent_nodes=8)), ("warm", dict(bnb_search=2, concurrent_nodes=8))):
        calls["n"] = 0
        r = L.BranchAndBound(bnb_mode=1, rank=rank, world=world, allreduce_max=allreduce_max, **kw).Solve(ps)
        res[name] = {"z": r.OptimalValue, "x": np.asarray(r.Solution).tolist() if r.Solution is not None else None,
                     "lp_solves": r.LpSolves, "nodes": r.Nodes, "allreduces": calls["n"], "aux": list(r.Aux)}
    # a mid-size 0/1 IP whose node tableaux (121 x 217 f64 = 205 KB) do not fit one CU's LDS, SOLVED by the sharded level search:
    # incumbents appear on both ranks, the all-reduced bound prunes, pools are rebalanced (tests/golden/bnb_mid.json has the
    # oracle's reference-order DFS optimum and HiGHS's)
    cm, Am, relm, bm = synth.binary_ip(96, 24)
    pm = L.LPProblem.from_arrays(0, cm, Am, relm, bm)
    calls["n"] = 0
    rm = L.BranchAndBound(bnb_mode=1, bnb_search=1, bnb_dive=1, concurrent_nodes=64, rank=rank, world=world, allreduce_max=allreduce_max).Solve(pm)
    logm = np.asarray(rm.NodeLog).reshape(-1, 3)
    res["mid"] = {"z": rm.OptimalValue, "x": np.asarray(rm.Solution).tolist(), "lp_solves": rm.LpSolves, "aux": list(rm.Aux), "allreduces": calls["n"],
                  "pruned": int((logm[:, 1] == 3).sum()), "incumbents": int((logm[:, 1] == 4).sum())}
    # knapsack solved to exhaustion through the device node store (evaluated tree re-seeded after the split)
    g = np.random.default_rng(3)
    n = 300
    w = g.integers(1, 60, size=n).astype(float); p = w + g.integers(0, 12, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    kp = L.LPProblem(L.Sense.Max, p.tolist(), [L.Constraint(w.tolist(), L.Rel.LE, cap)])
    calls["n"] = 0
    rk = L.BranchAndBoundKnapsack(max_nodes=0, concurrent_nodes=64, rank=rank, world=world, allreduce_max=allreduce_max).Solve(kp)
    res["knap"] = {"z": rk.OptimalValue, "x": rk.Extra.astype(int).tolist(), "popped": rk.Nodes, "allreduces": calls["n"]}
    # a failing rank in the WARM-started search (thousands of parked parents make LPX_ENOMEM the plausible failure there): rank 1's
    # 9th node group throws; both ranks must come back with an error after the same number of collectives
    calls["n"] = 0
    kw = {"test_fail_after_nodes": 9} if rank == 1 else {}
    try:
        L.BranchAndBound(bnb_mode=1, bnb_search=2, concurrent_nodes=4, rank=rank, world=world, allreduce_max=allreduce_max, **kw).Solve(ps)
        res["warm_failure"] = {"error": None, "allreduces": calls["n"]}
    except L.SolverException as e:
        res["warm_failure"] = {"error": str(e), "code": e.code, "allreduces": calls["n"]}
    json.dump(res, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
