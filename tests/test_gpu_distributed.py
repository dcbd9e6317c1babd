"""Sharded branch-and-bound and knapsack with the REAL device loops, and the streaming primal loop under GPU sharing: two ranks
share the visible GPU, the per-level exchange runs over gloo (the driver's multi-GPU runs use RCCL for the same callback).  Complements
tests/test_distributed_cpu.py, where the device loops are stood in for by the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from linear_programming_solver_lpr381_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_agree_with_the_single_rank_search(gpu, oracle, tmp_path):
    port = 29500 + (os.getpid() % 400)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PYTHONPATH=ROOT)
    outs = [str(tmp_path / f"r{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_gpu_worker.py"), str(r), "2", outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
    res = [json.load(open(o)) for o in outs]
    # the streaming primal loop under GPU sharing: the full 3000 pivots on both ranks, the same ones, the oracle's first 150
    c, A, b = synth.dense_lp(4096, 8192)
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    st_ref, tr_ref = oracle.primal_tableau(T, basis, max_iter=150)
    del T
    for r in res:
        assert r["primal"]["status"] == 3 and r["primal"]["pivots"] == 3000, r["primal"]
        assert r["primal"]["launches"] < 1.1 * 3000 + 200                  # the fused path: one launch per pivot
        assert r["primal"]["trace150"] == tr_ref.tolist()
    assert res[0]["primal"]["trace_sha"] == res[1]["primal"]["trace_sha"] and res[0]["primal"]["basis_sha"] == res[1]["primal"]["basis_sha"]
    # single-rank references on the same problems
    cs, As, rels, bs = synth.binary_ip(24, 8, seed=11)
    ps = gpu.LPProblem.from_arrays(0, cs, As, rels, bs)
    one = gpu.BranchAndBound(bnb_mode=1, bnb_search=1, bnb_dive=1, concurrent_nodes=8).Solve(ps)
    dfs = gpu.BranchAndBound(bnb_mode=1).Solve(ps)                       # the reference's own recursion
    assert one.OptimalValue == dfs.OptimalValue
    for name in ("cold", "warm"):
        a, b = res[0][name], res[1][name]
        assert a["z"] == b["z"], name                                     # both ranks end with the same incumbent ...
        if name == "cold":
            assert a["z"] == dfs.OptimalValue                             # ... the global optimum (same arithmetic as the DFS)
        else:                                                             # warm start: another pivot path, same optimum to 1e-9
            assert abs(a["z"] - dfs.OptimalValue) <= 1e-9 * abs(dfs.OptimalValue)
        assert a["x"] == b["x"], name                                     # ... and the same solution vector
        assert abs(np.asarray(a["x"]) @ cs - a["z"]) <= 1e-9 * max(1.0, abs(a["z"]))
        assert a["allreduces"] == b["allreduces"] >= 1 and a["aux"][0] == b["aux"][0]      # same levels, same collectives
    g = np.random.default_rng(3)
    n = 300
    w = g.integers(1, 60, size=n).astype(float); p = w + g.integers(0, 12, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=0)
    a, b = res[0]["knap"], res[1]["knap"]
    assert a["z"] == b["z"] == ref.best_z and a["x"] == b["x"]
    assert float(np.asarray(a["x"]) @ p) == ref.best_z and float(np.asarray(a["x"]) @ w) <= cap
    assert a["allreduces"] == b["allreduces"] >= 1
