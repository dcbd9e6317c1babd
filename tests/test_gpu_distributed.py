"""Sharded branch-and-bound and knapsack with the REAL device loops, and the streaming primal loop under GPU sharing: two ranks
share the visible GPU, the per-level exchange runs over gloo (the driver's multi-GPU runs use RCCL for the same callback).  Complements
tests/test_distributed_cpu.py, where the device loops are stood in for by the oracle."""
import hashlib
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
        so, se = p.communicate(timeout=900)
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
    # ---- the other launch types under sharing: bits of a solo run / of the oracle on both ranks ----
    _bits_sha = lambda a, dt=np.float64: hashlib.sha256(np.ascontiguousarray(a, dtype=dt).view(np.uint8)).hexdigest()
    # (a) revised loop at config-3 size, 300 iterations: equal across ranks and equal to a solo run here
    c3, A3, b3 = synth.dense_lp(4096, 8192)
    with gpu.DeviceRevised(A3, -c3, b3) as rv:
        status, st = rv.run(max_iter=300, batch=50)
        Bidx, Nidx, xB, z = rv.result()
        tr3 = rv.trace()
    del A3
    solo = {"status": int(status), "pivots": int(st["pivots"]), "trace_sha": _bits_sha(tr3, np.int32), "bidx_sha": _bits_sha(Bidx, np.int32),
            "nidx_sha": _bits_sha(Nidx, np.int32), "xb_sha": _bits_sha(xB), "z_hex": float(z).hex()}
    assert solo["status"] == 3 and solo["pivots"] == 300
    assert res[0]["revised"] == solo and res[1]["revised"] == solo
    # (b) dual streaming path on the 361 MB tableau: bitwise vs the oracle
    m, n = 4500, 5500
    cd, Ad, bd = synth.dense_lp(m, n, seed=11)
    Td, basd = synth.primal_tableau_from(cd, Ad, bd)
    del Ad
    g = np.random.Generator(np.random.PCG64(11))
    for i in g.choice(m, size=12, replace=False):
        Td[i, :n] *= -1.0
        Td[i, -1] = -0.02 * Td[i, -1]
    st_ref, tr_ref, nf = oracle.dual_tableau(Td, basd, fdf_guard=4, cleanup=1, max_iter=6)
    want = {"status": int(st_ref), "fdf": int(nf), "trace": np.asarray(tr_ref).tolist(), "basis_sha": _bits_sha(basd, np.int32), "T_sha": _bits_sha(Td)}
    del Td
    assert res[0]["dual"] == want and res[1]["dual"] == want
    # (c) forced pivots at 4097 x 12289 on the two-launch kernels: bitwise vs the oracle
    T0 = synth.raw_tableau(4097, 12289)
    rows, cols = synth.forced_pivot_list(4097, 12289, 8)
    chosen_ref = oracle.forced_pivots(T0, rows, cols, 0.1)
    want = {"chosen": np.asarray(chosen_ref).tolist(), "pivots": 8, "T_sha": _bits_sha(T0)}
    del T0
    assert res[0]["forced"] == want and res[1]["forced"] == want
    # (d) exact K7' at n = 600: bitwise vs the oracle's Invert
    gi = np.random.default_rng(6)
    M = gi.uniform(-1, 1, size=(600, 600))
    M[2, 0] = -M[1, 0]
    rc, inv_ref = oracle.invert(M)
    assert rc == 0 and res[0]["invert"]["inv_sha"] == res[1]["invert"]["inv_sha"] == _bits_sha(inv_ref)
    # a failing rank in the warm-started search: both ranks return an error, after the same number of collectives
    f0, f1 = res[0]["warm_failure"], res[1]["warm_failure"]
    assert f0["error"] and "peer rank failed" in f0["error"], f0
    assert f1["error"] and "injected failure" in f1["error"] and f1["code"] == -3, f1
    assert f0["allreduces"] == f1["allreduces"] >= 1
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
    # the mid-size IP (node tableaux larger than one CU's LDS): the optimum of the oracle's reference-order DFS (28 122 node LPs, a
    # minute of CPU: tests/golden/bnb_mid.json, made by tests/golden/gen_bnb_mid.py) and of HiGHS, on both ranks, with the shared
    # bound at work -- incumbents found, nodes pruned by it, the work split
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bnb_mid.json")))["96x24"]
    cm, Am, relm, bm = synth.binary_ip(96, 24)
    a, b = res[0]["mid"], res[1]["mid"]
    assert a["z"] == b["z"] == gold["dfs_z"] and abs(a["z"] - gold["highs_z"]) <= 1e-9 * gold["highs_z"]
    assert a["x"] == b["x"] and set(a["x"]) <= {0.0, 1.0}
    xm = np.asarray(a["x"])
    assert float(xm @ cm) == a["z"] and (Am @ xm <= bm + 1e-9).all()
    assert a["allreduces"] == b["allreduces"] >= 10 and a["aux"][0] == b["aux"][0]
    assert a["incumbents"] + b["incumbents"] >= 2 and a["pruned"] + b["pruned"] >= 1000
    assert min(a["lp_solves"], b["lp_solves"]) >= 0.25 * max(a["lp_solves"], b["lp_solves"])
    g = np.random.default_rng(3)
    n = 300
    w = g.integers(1, 60, size=n).astype(float); p = w + g.integers(0, 12, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=0)
    a, b = res[0]["knap"], res[1]["knap"]
    assert a["z"] == b["z"] == ref.best_z and a["x"] == b["x"]
    assert float(np.asarray(a["x"]) @ p) == ref.best_z and float(np.asarray(a["x"]) @ w) <= cap
    assert a["allreduces"] == b["allreduces"] >= 1


def _run(cmd, env=None, timeout=900):
    r = subprocess.run(cmd, env=dict(os.environ, PYTHONPATH=ROOT, **(env or {})), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return r


@pytest.mark.parametrize("mode", ["id", "tcp"])
def test_lpx_comm_world_of_one_on_the_real_rccl(oracle, tmp_path, mode):
    """X1 inside the library (include/lpx.h lpx_comm_*): ncclCommInitRank and ncclAllReduce(ncclMax, ncclDouble) execute -- at the
    only world size a 1-GPU box allows -- and the three sharded searches run their per-level / per-round exchange through them
    (Models/Branch&Bound.cs:182,191; Models/BranchAndBoundKnapsack.cs:124,157-160)."""
    out = str(tmp_path / "comm.json")
    port = 29900 + (os.getpid() % 90)
    _run([sys.executable, os.path.join(ROOT, "tests", "_comm_worker.py"), mode, out, str(port)])
    res = json.load(open(out))
    assert res["before"]["world"] == 0 and res["before"]["rank"] == -1
    assert res["info"]["world"] == 1 and res["info"]["rank"] == 0 and res["info"]["rccl_version"] > 20000
    assert res["roundtrip"] and res["roundtrip_big"]
    assert res["double_init"] and "exists already" in res["double_init"]
    for name in ("cold", "warm"):
        a, b = res[name], res[name]["plain"]
        assert a["z"] == b["z"] and a["x"] == b["x"] and a["lp_solves"] == b["lp_solves"], name
        assert a["aux"][0] == b["aux"][0]                                   # same levels ...
        assert a["rccl_allreduces"] == a["aux"][1] >= a["aux"][0] >= 2      # ... each with its all-reduce, + the publication of x
        assert b["aux"][1] == 0
    assert res["mismatch"] and "differ from the communicator" in res["mismatch"]
    g = np.random.default_rng(3)
    n = 300
    w = g.integers(1, 60, size=n).astype(float); p = w + g.integers(0, 12, size=n)
    cap = float(np.floor(0.5 * w.sum()))
    ref = oracle.knapsack_solve(oracle.Problem(oracle.MAX, p, w.reshape(1, -1), [oracle.LE], [cap]), max_nodes=0)
    assert res["knap"]["z"] == ref.best_z and res["knap"]["popped"] == ref.nodes_popped and res["knap"]["rccl_allreduces"] >= 2
    assert res["final"]["allreduce_ms"] > 0 and res["after"]["world"] == 0


def test_bench_runs_the_nccl_backend_with_a_world_of_one():
    """bench.py's multi-GPU plumbing on the one GPU at hand: init_process_group("nccl"), the id broadcast, lpx_comm_init and the
    searches' all-reduces over RCCL all execute (LPX_BENCH_FORCE_DIST=1 relaxes the `world > 1` guards)."""
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--pivots-per-step", "300",
              "--only", "bnb_prune,knapsack", "--knap-nodes", "20000"], env={"LPX_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(29700 + os.getpid() % 90)})
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["backend"] == "RCCL over xGMI"
    assert "lpx_comm" in d["config"]["engine_collective"] and d["config"]["lpx_comm"]["world"] == 1
    assert d["config"]["per_rank_pivots"] == [300.0]
    assert d["bnb_prune"]["allreduces"] >= d["bnb_prune"]["levels"] >= 2 and d["bnb_prune"]["incumbent"] is not None
    assert d["config"]["lpx_comm"]["allreduces"] >= d["bnb_prune"]["allreduces"] + 2
    assert d["knapsack"]["popped"] == 20000.0


def test_bench_launcher_runs_two_ranks_without_torchrun():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the two ranks before it touches the GPU; both
    share this box's GPU over the gloo rehearsal backend and each streams its own 403 MB LP (per-rank pivot counts on the line)."""
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras", "--steps", "1", "--warmup", "0",
              "--pivots-per-step", "600"], env={"LPX_BENCH_BACKEND": "gloo"})
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["per_rank_pivots"] == [600.0, 600.0] and d["config"]["rccl_ranks"] is None
    assert d["value"] > 0 and d["scaling"] == "weak"
