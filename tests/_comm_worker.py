"""Worker of tests/test_gpu_distributed.py::test_lpx_comm_*: a world of ONE on the real RCCL (the largest world a 1-GPU box can
host: RCCL refuses two ranks on one device).  ncclCommInitRank + ncclAllReduce(ncclMax, ncclDouble) run through liblpx's own
communicator (include/lpx.h lpx_comm_*), and with LPX_COMM_SHARD_ONE=1 the sharded code path of the three searches -- hand-out,
one all-reduce per level / round, publication of x -- runs over it.  In a process of its own so that RCCL never lives in pytest's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LPX_COMM_SHARD_ONE"] = "1"

import numpy as np

import linear_programming_solver_lpr381_amd as L
from linear_programming_solver_lpr381_amd import synth


def main():
    mode, out = sys.argv[1], sys.argv[2]
    L._lib.check(L._lib.lib().lpx_init(0))
    res = {"before": L.comm.info()}
    if mode == "tcp":
        L.comm.init_tcp(0, 1, "127.0.0.1", int(sys.argv[3]))
    else:
        uid = L.comm.unique_id()
        res["id_nonzero"] = any(uid)
        L.comm.init(0, 1, uid)
    res["info"] = L.comm.info()
    v = np.array([3.0, -np.inf, 5.5, -0.0, 1e300])
    res["roundtrip"] = bool(np.array_equal(L.comm.allreduce_max(v).view(np.uint64), v.view(np.uint64)))
    big = np.arange(100_000, dtype=np.float64) - 5e4            # larger than the first staging buffer: it grows
    res["roundtrip_big"] = bool(np.array_equal(L.comm.allreduce_max(big), big))
    try:
        L.comm.init(0, 1, L.comm.unique_id())
        res["double_init"] = None
    except L.LpxError as e:
        res["double_init"] = str(e)
    # the sharded searches over the communicator (no allreduce_max callback): same results as the unsharded ones
    cs, As, rels, bs = synth.binary_ip(24, 8, seed=11)
    ps = L.LPProblem.from_arrays(0, cs, As, rels, bs)
    L.comm.destroy()
    plain = {}
    for name, kw in (("cold", dict(bnb_search=1, bnb_dive=1, concurrent_nodes=8)), ("warm", dict(bnb_search=2, concurrent_nodes=8))):
        r = L.BranchAndBound(bnb_mode=1, **kw).Solve(ps)
        plain[name] = {"z": r.OptimalValue, "x": np.asarray(r.Solution).tolist(), "lp_solves": r.LpSolves, "aux": list(r.Aux)}
    L.comm.init(0, 1, L.comm.unique_id())
    n0 = L.comm.info()["allreduces"]
    for name, kw in (("cold", dict(bnb_search=1, bnb_dive=1, concurrent_nodes=8)), ("warm", dict(bnb_search=2, concurrent_nodes=8))):
        r = L.BranchAndBound(bnb_mode=1, rank=0, world=1, **kw).Solve(ps)
        n1 = L.comm.info()["allreduces"]
        res[name] = {"z": r.OptimalValue, "x": np.asarray(r.Solution).tolist(), "lp_solves": r.LpSolves, "aux": list(r.Aux),
                     "rccl_allreduces": n1 - n0, "plain": plain[name]}
        n0 = n1
    try:                                                        # opts that disagree with the communicator are refused
        L.BranchAndBound(bnb_mode=1, bnb_search=1, rank=1, world=2).Solve(ps)
        res["mismatch"] = None
    except L.SolverException as e:
        res["mismatch"] = str(e)
    g = np.random.default_rng(3)
    n = 300
    w = g.integers(1, 60, size=n).astype(float); p = w + g.integers(0, 12, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    kp = L.LPProblem(L.Sense.Max, p.tolist(), [L.Constraint(w.tolist(), L.Rel.LE, cap)])
    rk = L.BranchAndBoundKnapsack(max_nodes=0, concurrent_nodes=64, rank=0, world=1).Solve(kp)
    res["knap"] = {"z": rk.OptimalValue, "x": rk.Extra.astype(int).tolist(), "popped": rk.Nodes,
                   "rccl_allreduces": L.comm.info()["allreduces"] - n0}
    res["final"] = L.comm.info()
    L.comm.destroy()
    res["after"] = L.comm.info()
    json.dump(res, open(out, "w"))


if __name__ == "__main__":
    main()
