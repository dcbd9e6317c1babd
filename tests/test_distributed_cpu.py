"""CPU, world_size 2, gloo: the N>1 path (frontier sharding + one all-reduce(max) per level / round +
termination + final ownership of x) of the product's host logic.  See tests/_dist_worker.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_world2_gloo_bnb_and_knapsack(tmp_path, lpx, oracle):
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs, outs = [], []
    for r in range(2):
        out = str(tmp_path / f"r{r}.json")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), "2", out], env=env))
    for p in procs:
        assert p.wait(timeout=280) == 0
    res = [json.load(open(o)) for o in outs]
    for r in res:
        # same optimum as the reference-order DFS of the oracle, on every rank
        assert r["bnb"]["z"] == r["bnb_ref"]["z"]
        assert r["knap"]["z"] == r["knap_ref"]["z"]
        assert r["knap"]["feasible"] and r["knap"]["value"] == r["knap"]["z"]
    # both ranks end with the same published solution vectors
    assert res[0]["bnb"]["x"] == res[1]["bnb"]["x"]
    assert res[0]["knap"]["x"] == res[1]["knap"]["x"]
    # the collective really ran, the same number of times on both ranks (no rank left a level early)
    assert res[0]["bnb"]["allreduces"] == res[1]["bnb"]["allreduces"] >= 3
    assert res[0]["knap"]["allreduces"] == res[1]["knap"]["allreduces"] >= 2
    # the work was actually split: neither rank solved every node alone
    tot = res[0]["bnb"]["lp_solves"] + res[1]["bnb"]["lp_solves"]
    assert res[0]["bnb"]["lp_solves"] < tot and res[1]["bnb"]["lp_solves"] < tot
    # whole-job accounting: the replicated warm-up is counted once (by rank 0), so the sum over ranks cannot
    # exceed what a single process visiting every node would count by more than the pruning differences allow
    single = res[0]["bnb_single"]
    assert single["z"] == res[0]["bnb"]["z"]
    assert tot <= 2 * single["lp_solves"] + 8
    # tiny GLOBAL node budgets (1..12): both ranks stop together -- same number of collectives, no hang -- and the
    # job as a whole stays within the budget (+ world - 1 for the rounding of the split)
    for b0, b1 in zip(res[0]["budget"], res[1]["budget"]):
        assert b0["cap"] == b1["cap"] and b0["allreduces"] == b1["allreduces"], (b0, b1)
        assert b0["nodes"] + b1["nodes"] <= b0["cap"] + 1, (b0, b1)
    # equal-z ties: both ranks publish the same x, it is the x of the single-process search, and its objective is the DFS optimum
    t0, t1 = res[0]["ties"], res[1]["ties"]
    assert t0["z"] == t1["z"] == t0["single_z"] == t0["dfs_z"]
    assert t0["x"] == t1["x"] == t0["single_x"] == t0["dfs_x"]      # ... and the vector the reference's depth-first order finds first
    # rebalancing: with one node per round and rank the pools drift apart; descriptors moved, every rank kept working,
    # and the optimum is unchanged
    r0, r1 = res[0]["rebalance"], res[1]["rebalance"]
    assert r0["z"] == r1["z"] == res[0]["bnb_ref"]["z"] and r0["x"] == r1["x"]
    assert r0["allreduces"] == r1["allreduces"]
    assert r0["aux"][2] >= 1 and r0["aux"][3] >= 1 and r0["aux"][2:] == r1["aux"][2:]      # same plan on both ranks
    assert min(r0["lp_solves"], r1["lp_solves"]) >= 0.25 * max(r0["lp_solves"], r1["lp_solves"])


def test_single_process_level_search_with_seam_matches_oracle(lpx, oracle):
    import numpy as np
    sys.path.insert(0, HERE)
    from _dist_worker import node_lp
    g = np.random.default_rng(5)
    n, m = 9, 4
    A = g.integers(0, 10, size=(m, n)).astype(float)
    b = np.floor(0.5 * A.sum(axis=1))
    c = g.integers(1, 21, size=n).astype(float)
    Af = np.vstack([A, np.eye(n)]); bf = np.concatenate([b, np.ones(n)])
    p = lpx.LPProblem.from_arrays(0, c, Af, np.zeros(m + n, int), bf)
    for mode in (0, 1):
        ref = oracle.bnb_solve(oracle.Problem(oracle.MAX, c, Af, np.zeros(m + n, np.int32), bf), mode)
        dfs = lpx.BranchAndBound(bnb_mode=mode, bnb_search=0, test_node_lp=node_lp).Solve(p)
        assert dfs.NodeLog.tolist() == ref.log.tolist() and dfs.LpSolves == ref.lp_solves
        lvl = lpx.BranchAndBound(bnb_mode=mode, bnb_search=1, concurrent_nodes=3, test_node_lp=node_lp).Solve(p)
        dive = lpx.BranchAndBound(bnb_mode=mode, bnb_search=1, bnb_dive=1, concurrent_nodes=2, test_node_lp=node_lp).Solve(p)
        if ref.has_incumbent:
            assert dfs.OptimalValue == ref.best_z
            if mode == 1:
                assert lvl.OptimalValue == ref.best_z and dive.OptimalValue == ref.best_z
