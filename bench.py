#!/usr/bin/env python3
"""bench.py -- headline benchmark of the lpx simplex / branch-and-bound hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

value (pivots/s)   The HBM-streaming workload: the dense random LP m=4096 n=8192 (the size BASELINE.json's
                   configs[2] names; north_star's "4096x8192 dense tableau") solved by the primal tableau
                   simplex.  Its tableau is 4097 x 12289 f64 = 403 MB -- larger than the 256 MiB Infinity
                   Cache, so every pivot is a real HBM stream (the raw 4096x8192 shape is EXACTLY 256 MiB and
                   is reported beside it as `roofline_north_star`).  A "step" restores the tableau from a
                   pristine HBM snapshot (D2D, inside the timed region) and runs the first `--pivots-per-step`
                   pivots of the solve from the slack basis -- default 10000, the reference's own iteration cap
                   (Models/PrimalSimplex.cs:54,95-96; the full solve needs 80477 pivots) -- on the streaming
                   kernel lpx_pivot_fused (one launch per pivot: update(k) out of place beside select(k+1)).
                   value = pivots of all ranks / max wall.
                   A single LP does not shard (DESIGN.md "Multi-GPU"): with N ranks each rank solves its own
                   replica on its own GPU ("replicas only", weak scaling).
Extra objects on the same JSON line (rank 0 unless stated):
  roofline             lpx_pivot_fused (the rank-1 pivot update, Models/PrimalSimplex.cs:251-256, with the next
                       pivot's selection riding in 32 of its 33 158 workgroups) in THAT solve:
                       16*R*C algorithmic bytes per launch / mean launch duration from HIP events bound to each
                       dispatch on the library's stream (hipExtLaunchKernelGGL start/stop events), measured live.
                       `traffic` = HBM bytes per launch from the committed rocprofv3 --pmc passes over the same
                       workload (profiles/, see `traffic_source`); PMC needs the profiler around the process.
  roofline_north_star  the same kernel on the raw 4096x8192 tableau (forced pivots), 268 MB = the MALL size.
  config2              BASELINE configs[1] (m=1024 n=2048): the LDS-resident persistent kernel (latency bound,
                       no HBM roofline) and the streaming kernels on the same LP.
  cpu_baseline         CPU oracle (C port of the reference's scalar loops), 1 core, on a bounded sample of the
                       SAME 4097x12289 LP; plus an all-cores OpenMP courtesy figure and samples of the other legs.
  revised              config 3: revised simplex m=4096 n=8192, iterations/s over a bounded run.
  bnb / bnb_warm       config 4: 0/1 IP n=512 m=256 (+512 bound rows), repaired mode, node queue SHARDED over all
                       ranks (one all-reduce(max) per level); nodes/s = LP relaxations of all ranks / max wall.
  knapsack             config 5: 100k-item 0/1 knapsack, best-first B&B with batched GPU bounds, sharded subtrees.
"""
import argparse
import json
import os
os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # cpu_baseline's OpenMP team must not spin
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "simplex pivots/sec on dense m×n tableau; B&B nodes/sec at 1/2/4/8 GPUs"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
PROFILE_ROUND = "r03"


def update_kernel(R, C):
    """(kernel name, grid size in threads) of the rank-1 update for an R x C tableau -- the key of the committed profile
    summaries.  Mirrors launch_update_mb (csrc/lpx_kernels.hip): tableaux above 292 MiB take a streaming variant (one wave
    per workgroup, 3 rows per wave, non-temporal loads, one row in three of every ceil(bytes / 768 MiB)-th row block stored
    through the Infinity Cache: `_m`; `_s`, non-temporal stores throughout, only beyond 8 GiB); smaller ones the 8-row /
    256-lane kernel."""
    ld = (C + 15) // 16 * 16
    if 8 * ld * R > (292 << 20):
        name = "lpx::lpx_update_mb_m" if 8 * ld * R <= (8192 << 20) else "lpx::lpx_update_mb_s"
        return name, ((ld + 127) // 128) * ((R + 2) // 3) * 64
    units = ((ld + 127) // 128) * ((R + 7) // 8)
    return "lpx::lpx_update_mb", ((units + 3) // 4) * 256


def pivot_kernel(R, C):
    """The kernel that streams the tableau in the PRIMAL loop (no per-pivot callback): lpx_pivot_fused -- update(k) out of
    place (second tableau buffer) with select(k+1) in the first min(32, ceil(C/256)) workgroups of the same grid, 256-lane
    workgroups of four update waves; `_c` (default cache policy) while both buffers share the Infinity Cache (<= 152 MiB
    each), nontemporal loads + the mixed store policy above (run_fused / fused_policy, csrc).  LPX_FUSED_PIVOT=0 restores the
    two-launch in-place path: then the update kernel above."""
    ld = (C + 15) // 16 * 16
    if os.environ.get("LPX_FUSED_PIVOT", "1")[:1] == "0":
        return update_kernel(R, C)
    units = ((ld + 127) // 128) * ((R + 2) // 3)
    name = "lpx::lpx_pivot_fused_c" if 8 * ld * R <= (152 << 20) else "lpx::lpx_pivot_fused"
    return name, (min(32, (C + 255) // 256) + (units + 3) // 4) * 256


def committed_profile(name):
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{name}")
    return (json.load(open(path)), os.path.relpath(path, ROOT)) if os.path.exists(path) else (None, None)


def rocprof_kernel_us(R, C, primal=False):
    """Mean duration of the update kernel at this shape in the committed rocprofv3 --kernel-trace of this script
    (tools/trace_by_shape.py) -- a cross-check of the live HIP-event figure, NOT a measurement of this run."""
    d, _ = committed_profile("kernel_by_shape.json")
    kernel, grid = pivot_kernel(R, C) if primal else update_kernel(R, C)
    e = d.get(f"{kernel}@grid{grid}x1") if d else None
    return e["mean_ns_live"] / 1e3 if e else None


def pmc_traffic(R, C, primal=False):
    """HBM bytes per launch of the tableau-streaming kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    in separate runs, FETCH doubled per the gfx950 correction of MI355X_MICROARCH.md; tools/pmc_summarise.py)."""
    d, src = committed_profile("pmc_traffic.json")
    if d:
        kernel, grid = pivot_kernel(R, C) if primal else update_kernel(R, C)
        v = d.get(f"{kernel}@grid{grid}")
        if v:
            return v["hbm_bytes_per_launch"], src
    return None, None


def progress(msg):
    """One line per leg on stderr: the GPU pool kills runs that stay silent for minutes."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the launcher.  It has not imported torch, numpy
    or the package and has made no HIP call (a process that has initialised the GPU must never be replaced or forked from): it
    starts N children of this script, one rank per GPU, with the environment torch.distributed.run would give them, relays rank
    0's JSON line (the children share this stdout / stderr) and exits non-zero if any child did."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [None] * n
    t_fail = None
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
        if t_fail is None and any(rc not in (None, 0) for rc in rcs):
            t_fail = time.monotonic()                      # a rank died: its peers may sit in a collective for good
        if t_fail is not None and time.monotonic() - t_fail > 60.0:
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.kill()
        time.sleep(0.2)
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"[bench] ranks failed (rank, exit code): {bad}", file=sys.stderr, flush=True)
        sys.exit(1)
    sys.exit(0)


WARM_NOTE = ("one launch per step for the whole batch (lpx_group_fused: update out of place beside the select of every live node, the live "
             "list compacted on the device from launch to launch); two rolling batches alternate (lpx_multi_run_begin / _end), so the host "
             "reads back, parks and refills one while the other pivots")
LEGS = ("bnb", "bnb_warm", "bnb_prune", "bnb_prune_mid", "knapsack", "roofline", "config2", "revised", "cpu")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--launch-check", action="store_true", help="every rank prints its launch environment as JSON and exits")
    ap.add_argument("--only", default="", help="comma-separated subset of the extra legs: " + ",".join(LEGS))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--m", type=int, default=4096)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--pivots-per-step", type=int, default=10000,
                    help="pivots of the solve per step (default: the reference's iteration cap)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline value")
    ap.add_argument("--roofline-pivots", type=int, default=400)
    ap.add_argument("--headline-pivots", type=int, default=200)
    ap.add_argument("--cpu-sample-pivots", type=int, default=240)
    ap.add_argument("--bnb-nodes", type=int, default=3200, help="GLOBAL node budget of the config 4 leg (strong scaling)")
    ap.add_argument("--bnb-prune-n", type=int, default=60)
    ap.add_argument("--bnb-prune-m", type=int, default=12)
    ap.add_argument("--bnb-concurrent", type=int, default=256)
    ap.add_argument("--bnb-mid-n", type=int, default=128)
    ap.add_argument("--bnb-mid-m", type=int, default=32)
    ap.add_argument("--bnb-mid-seed", type=int, default=20251003)
    ap.add_argument("--bnb-mid-concurrent", type=int, default=256)
    ap.add_argument("--bnb-mid-nodes", type=int, default=0, help="GLOBAL node budget of the mid-size leg (0 = to optimality)")
    ap.add_argument("--bnb-warm-nodes", type=int, default=8000, help="GLOBAL node budget of the warm-start leg")
    ap.add_argument("--bnb-warm-concurrent", type=int, default=256)
    ap.add_argument("--knap-nodes", type=int, default=1000000, help="pop budget per rank (config 5 leg)")
    ap.add_argument("--revised-iters", type=int, default=300)
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    if only - set(LEGS):
        ap.error(f"--only: unknown leg(s) {sorted(only - set(LEGS))}")

    def leg(name):
        return not args.no_extras and (not only or name in only)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus)                            # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a 1-GPU number must never pass for an N-GPU one (or the reverse)
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or without a "
              "launcher (bench.py then starts the ranks itself)", file=sys.stderr, flush=True)
        sys.exit(2)

    if args.launch_check:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master_addr": os.environ.get("MASTER_ADDR"),
                          "master_port": os.environ.get("MASTER_PORT")}), flush=True)
        return

    import numpy as np
    import torch
    import torch.distributed as dist

    import linear_programming_solver_lpr381_amd as L
    from linear_programming_solver_lpr381_amd import synth

    # LPX_BENCH_BACKEND=gloo is a REHEARSAL switch: several ranks share the GPUs that are visible (rank r
    # uses device r % device_count) and the collectives run over gloo on CPU tensors.  The driver's runs
    # use the default: one rank per GPU, RCCL ("nccl").
    # LPX_BENCH_FORCE_DIST=1 (test knob): a world of ONE still initialises the process group and the library's
    # communicator and runs the sharded code path of the searches, so that init_process_group("nccl"), lpx_comm_init and
    # the per-level ncclAllReduce have executed on a 1-GPU box before an 8-GPU node sees them.
    backend = os.environ.get("LPX_BENCH_BACKEND", "nccl")
    force_dist = os.environ.get("LPX_BENCH_FORCE_DIST", "0")[:1] == "1"
    use_dist = world > 1 or force_dist
    if force_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ["LPX_COMM_SHARD_ONE"] = "1"            # read by liblpx when the first lpx_solve runs
    backend_name = "RCCL over xGMI" if backend == "nccl" else f"{backend} (rehearsal backend, CPU tensors)"
    dev = local_rank % max(torch.cuda.device_count(), 1) if backend == "gloo" else local_rank
    torch.cuda.set_device(dev)
    L._lib.check(L._lib.lib().lpx_init(dev))
    coll_dev = "cpu" if backend == "gloo" else "cuda"
    rccl_ranks = None
    engine_comm = None           # how the searches exchange their bound
    if use_dist:
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            rccl_ranks = dist.get_world_size()
            # the ENGINE's communicator (liblpx's own RCCL, include/lpx.h lpx_comm_*): rank 0 makes the id, torch ships it
            idt = torch.zeros(L.comm.ID_BYTES, dtype=torch.uint8, device="cuda")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(L.comm.unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, src=0)
            comm_err = None
            try:
                L.comm.init(rank, world, bytes(idt.cpu().numpy().tobytes()))
                chk = L.comm.allreduce_max(np.array([float(rank), -float(rank)]))
                if chk.tolist() != [float(world - 1), 0.0]:
                    comm_err = f"lpx_comm all-reduce(max) returned {chk.tolist()}"
            except Exception as e:                         # noqa: BLE001 -- any failure here must not cost the run its numbers
                comm_err = f"{type(e).__name__}: {e}"
            # every rank takes the same path: one torch all-reduce says whether anybody's communicator failed
            okt = torch.tensor([0.0 if comm_err else 1.0], device="cuda")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            if okt.item() >= 1.0:
                engine_comm = "lpx_comm"
            else:
                # the searches then exchange their bound through the host callback over torch.distributed (same RCCL, torch's
                # communicator); the line says so (`engine_collective`)
                print(f"[bench] rank {rank}: liblpx's communicator is not used ({comm_err or 'a peer failed'}); "
                      "falling back to the torch.distributed callback", file=sys.stderr, flush=True)
                if not comm_err:
                    L.comm.destroy()
                engine_comm = None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce_max_torch(vals):
        """X1 through torch.distributed (the gloo rehearsal's path: the library's communicator is RCCL only)."""
        t = torch.from_numpy(np.ascontiguousarray(vals)).to(coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().numpy()

    # what the sharded searches get as `allreduce_max`: None = the library's communicator (or one process)
    allreduce_max = None if (engine_comm == "lpx_comm" or not use_dist) else allreduce_max_torch

    def reduce_sum_max(count, seconds):
        if not use_dist:
            return float(count), float(seconds)
        a = torch.tensor([float(count)], dtype=torch.float64, device=coll_dev)
        b = torch.tensor([float(seconds)], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM)
        dist.all_reduce(b, op=dist.ReduceOp.MAX)
        return float(a.item()), float(b.item())

    def gather_list(v):
        """per-rank values (list over ranks)"""
        if not use_dist:
            return [float(v)]
        t = torch.tensor([float(v)], dtype=torch.float64, device=coll_dev)
        outl = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outl, t)
        return [float(x.item()) for x in outl]

    # ---- headline workload: primal tableau simplex on the m=4096 n=8192 LP (tableau 403 MB) -------
    m, n = args.m, args.n
    c, A, b = synth.dense_lp(m, n, seed=synth.SEED + rank)   # one replica per rank, own seed
    T, basis = synth.primal_tableau_from(c, A, b)
    del A
    R, C = T.shape
    dt = L.DeviceTableau.from_host(T, basis)
    dt.snapshot()
    opts = L.default_opts(False, batch=args.batch, use_graph=0 if args.no_graph else 1, max_iter=args.pivots_per_step)

    def step():
        dt.restore()
        status, st = dt.primal_run(opts)
        assert status in (0, 3), f"solve ended with status {status}"     # OPTIMAL, or the iteration cap of the step
        # a step that stopped at the cap has made exactly that many pivots; the default LP of rank 0 needs 80 477, so a
        # shorter run there is a wrong run, not a fast one (the other ranks' LPs -- other seeds -- are only held to the first rule)
        assert status == 0 or st["pivots"] == args.pivots_per_step, f"cap reached after {st['pivots']} pivots"
        assert not (rank == 0 and (m, n) == (4096, 8192) and args.pivots_per_step <= 50000) or st["pivots"] == args.pivots_per_step, \
            f"the headline LP stopped after {st['pivots']} pivots with status {status}"
        return st

    progress(f"headline solves ({R}x{C}, {args.pivots_per_step} pivots per step)")
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    pivots = 0
    loop_ms = 0.0
    for _ in range(args.steps):
        st = step()
        pivots += st["pivots"]
        loop_ms += st["loop_ms"]
    barrier()
    dt_s = time.perf_counter() - t0
    total_pivots, wall = reduce_sum_max(pivots, dt_s)
    per_rank_pivots = gather_list(pivots)

    out = {
        "metric": METRIC,
        "value": total_pivots / wall,
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"dense random LP m={m} n={n}, primal tableau simplex, tableau {R}x{C} f64 "
                        f"({R * C * 8 / 1e6:.0f} MB > 256 MiB Infinity Cache), first {args.pivots_per_step} pivots "
                        "from the slack basis per step (the reference's iteration cap, Models/PrimalSimplex.cs:54)",
            "parallelism": "replicas only (one LP per GPU)" if world > 1 else "1 GPU",
            "backend": backend_name if use_dist else "none (1 process)",
            "rccl_ranks": rccl_ranks,
            "engine_collective": ("liblpx's own RCCL communicator (lpx_comm_*: ncclAllReduce(ncclMax, ncclDouble), one per level / round)"
                                  if engine_comm == "lpx_comm" else
                                  "host callback -> torch.distributed all_reduce" if use_dist else "none (1 process)"),
            "pivots_per_step": pivots / max(args.steps, 1),
            "per_rank_pivots": per_rank_pivots,
            "batch": args.batch,
            "hipgraph": not args.no_graph,
            "device_loop_ms_per_step": loop_ms / max(args.steps, 1),
            "path": (("streaming, one launch per pivot (lpx_pivot_fused: update(k) out of place beside select(k+1))"
                      if st["launches"] < 1.5 * max(st["pivots"], 1) else "streaming (lpx_select_mb + lpx_update_mb per pivot)")
                     if st["launches"] > 2 else "resident (tableau in LDS, 1 launch per solve)"),
            "launches_per_step": st["launches"],
            "restore_inside_timed_region": True,
        },
    }

    if not args.no_extras:
        progress(f"value leg done ({dt_s:.1f} s); B&B leg (config 4)")
        # ---- config 4: sharded branch and bound (all ranks), STRONG scaling: one global node budget ---------
        def gather_counts(v):
            """per-rank values (list over ranks), via all_gather"""
            if not use_dist:
                return [float(v)]
            t = torch.tensor([float(v)], dtype=torch.float64, device=coll_dev)
            outl = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(outl, t)
            return [float(x.item()) for x in outl]

        def bnb_leg(problem, label, warm_full=False, **kw):
            if args.warmup > 0:
                # untimed warm-up, as the headline's W steps: a short search with the same pool width, so that the handle
                # pools, pinned staging and graph captures of this mode exist before the clock starts (a first search in
                # a process measured 5.8 k warm nodes/s against 6.9-7.0 k for every later one, tools/probe_warm.py).
                # warm_full: the SAME search once, untimed -- the warm-started mode parks parent tableaux in a store that
                # grows by 1 GB hipMallocs; they cost the leg up to half its time when the bench followed other GPU work on
                # the box (4.5-4.8 k nodes/s instead of 6.8-7.2 k).  The library keeps a destroyed store's chunks for the
                # next one, so after this warm-up the timed search allocates nothing.
                wkw = dict(kw) if warm_full else dict(kw, max_nodes=2 * kw.get("concurrent_nodes", 64))
                L.BranchAndBound(bnb_mode=1, rank=rank, world=world, allreduce_max=allreduce_max, **wkw).Solve(problem)
            solver = L.BranchAndBound(bnb_mode=1, rank=rank, world=world, allreduce_max=allreduce_max, **kw)
            ar0 = L.comm.info()["allreduce_ms"] if engine_comm else 0.0
            barrier()
            t1 = time.perf_counter()
            rb = solver.Solve(problem)
            barrier()
            tb = time.perf_counter() - t1
            ar_ms = L.comm.info()["allreduce_ms"] - ar0 if engine_comm else None
            lp_total, tb_max = reduce_sum_max(rb.LpSolves, tb)
            piv_total, _ = reduce_sum_max(rb.Stats["pivots"], tb)
            per_rank = gather_counts(rb.LpSolves)
            aux = list(rb.Aux) if rb.Aux else [0, 0, 0, 0]
            log = np.asarray(rb.NodeLog).reshape(-1, 3) if rb.NodeLog is not None and len(rb.NodeLog) else np.zeros((0, 3), int)
            pruned, _ = reduce_sum_max(int((log[:, 1] == 3).sum()), tb)          # outcome 3 = pruned by bound, 4 = new incumbent (bnb.cpp)
            incs, _ = reduce_sum_max(int((log[:, 1] == 4).sum()), tb)
            return {"workload": label, "_pruned": pruned, "_incumbents": incs, "nodes_per_s": lp_total / tb_max, "lp_relaxations": lp_total, "pivots": piv_total,
                    "pivots_per_node": piv_total / max(lp_total, 1), "wall_s": tb_max,
                    "incumbent": rb.OptimalValue if rb.OptimalValue > -1e300 else None,
                    "scaling": "strong (one global node budget, split over the ranks at the hand-out)",
                    "warmup": ("none" if args.warmup <= 0 else "the same search once, untimed" if warm_full
                               else "one untimed search of 2 x the pool width in nodes"),
                    "per_rank_lp_relaxations": per_rank,
                    "imbalance_max_over_mean": max(per_rank) / max(sum(per_rank) / len(per_rank), 1e-9),
                    "levels": aux[0], "allreduces": aux[1], "allreduce_ms_rank0": ar_ms, "rebalancing_rounds": aux[2], "node_descriptors_moved": aux[3],
                    "collective": f"1 all-reduce(max) of {{incumbent, have_work, failed, max depth, pool size per rank}} per level "
                                  f"(+1 when descriptors move), {backend_name}" + (", through liblpx's communicator" if engine_comm else ", through the host callback")
                                  if use_dist else "none (1 rank)"}

        # what the node evaluation of config 4 amounts to in the path's unit (16*R*C bytes per pivot); the root
        # tableau's shape is a lower bound of every node's (a node at depth d has d more rows and columns)
        R0, C0 = 256 + 512 + 1, 512 + 256 + 512 + 1

        def hbm_equivalent(res):
            return res["pivots"] * 16.0 * R0 * C0 / res["wall_s"] / 1e9

        if leg("bnb") or leg("bnb_warm") or leg("cpu"):
            cb, Ab, relb, bb = synth.binary_ip(512, 256)
            pb = L.LPProblem.from_arrays(0, cb, Ab, relb, bb)
        if leg("bnb"):
            res = bnb_leg(pb, "random 0/1 IP n=512 m=256 + 512 rows x_j<=1 (config 4), repaired mode, level-synchronous "
                              f"sharded node queue, every node re-solved from the slack basis as the reference does; GLOBAL budget "
                              f"{args.bnb_nodes} nodes, {args.bnb_concurrent} node LPs in flight per GPU",
                          bnb_search=1, concurrent_nodes=args.bnb_concurrent, max_nodes=args.bnb_nodes)
            res["incumbent_note"] = ("no integer node exists within the reference's recursion cap: the LP relaxation has up to 256 "
                                     "fractional basic variables, one is fixed per level and SolveNode stops at depth 200 "
                                     "(Models/Branch&Bound.cs:25,132) -- a depth-first-K dive of 4000 nodes ends in 'maximum depth' "
                                     "leaves (tools/probe_bnb.py); the shared bound is exercised by `bnb_prune` / `bnb_prune_mid` below")
            # NOT a roofline fraction: the node tableaux of this leg live on chip (registers / LDS), HBM sees each once per launch
            res["hbm_equivalent"] = {"unit": "GB/s", "value": hbm_equivalent(res), "bound": "latency",
                                     "basis": f"pivots x 16*{R0}*{C0} B (root shape: a lower bound) / whole-leg wall time, host work included",
                                     "note": "the rate a streaming implementation would have to sustain to match this leg; the node tableaux live on "
                                             "chip (lpx_resident_group_r: seven 7.9 MB nodes at a time in the register files; lpx_resident_group: "
                                             "four in LDS), so this is not HBM traffic and no fraction of the HBM peak"}
            out["bnb"] = res
        # ---- config 4 again with warm-started children (NOT the reference's re-solve) ------
        if leg("bnb_warm"):
            res = bnb_leg(pb, "config 4, same sharded level search, children warm-started from the parent's final tableau (dual "
                              f"loop only) -- an engine mode, not the reference's algorithm; GLOBAL budget {args.bnb_warm_nodes} nodes, "
                              f"{args.bnb_warm_concurrent} node LPs per group call; node LPs on the register + LDS resident group kernel "
                              "(twelve 7.9 MB nodes on chip, a group is one launch)",
                          warm_full=True, bnb_search=2, concurrent_nodes=args.bnb_warm_concurrent, max_nodes=args.bnb_warm_nodes)
            res["hbm_equivalent"] = {"unit": "GB/s", "rate": hbm_equivalent(res),
                                     "note": "pivots x 16*R*C / wall: what a streaming engine would have to move -- the tableaux live in registers and LDS "
                                             "here, HBM sees a node twice (load, store): a latency-bound kernel, not a roofline fraction"}
            # the same search on the streaming form (LPX_WARM_RESIDENT=0, read per solve): 64 nodes per rolling batch through lpx_group_fused, HBM bound --
            # the path of node LPs wider than the register kernel's 1536 columns, and the leg's `roofline` of rounds 2 and 3
            os.environ["LPX_WARM_RESIDENT"] = "0"
            try:
                res_s = bnb_leg(pb, "the same search on the streaming kernels: two rolling batches of 64 node LPs through lpx_group_fused",
                                warm_full=True, bnb_search=2, concurrent_nodes=64, max_nodes=args.bnb_warm_nodes)
            finally:
                del os.environ["LPX_WARM_RESIDENT"]
            rate = hbm_equivalent(res_s)
            res["streaming"] = {"workload": res_s["workload"], "nodes_per_s": res_s["nodes_per_s"], "lp_relaxations": res_s["lp_relaxations"],
                                "pivots": res_s["pivots"], "wall_s": res_s["wall_s"],
                                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": rate, "frac": rate / HBM_PEAK_GBS,
                                             "basis": f"pivots x 16*{R0}*{C0} B (root shape: a lower bound) / whole-leg wall time, host work included",
                                             "note": WARM_NOTE}}
            # rounds 2 and 3 quoted this leg's whole-leg HBM fraction as `bnb_warm.roofline`: it is the STREAMING form's figure (the default
            # form above moves no tableau through HBM per pivot, so it has no HBM roofline)
            res["roofline"] = dict(res["streaming"]["roofline"], form="streaming kernels (bnb_warm.streaming), not the default resident form",
                                   nodes_per_s=res_s["nodes_per_s"])
            if rank == 0:
                # the leg's kernel by itself: lpx_group_fused on 64 copies of the root tableau pivoting in lock step (every slot live), HIP
                # events bound to each launch; traffic = the committed PMC passes of the same launch shape (tools/k4_headline.py)
                Tb_, basb_ = synth.primal_tableau_from(cb, Ab, bb)
                K_ = 64
                nodes_ = [L.DeviceTableau.from_host(Tb_, basb_) for _ in range(K_)]
                L.multi_run(nodes_, [False] * K_, L.default_opts(False, max_iter=64, resident=-1), L.default_opts(True, resident=-1))
                for t_ in nodes_:
                    t_.upload(Tb_, basb_)
                _, gss = L.multi_run(nodes_, [False] * K_, L.default_opts(False, max_iter=192, resident=-1, profile=1),
                                     L.default_opts(True, resident=-1, profile=1))
                for t_ in nodes_:
                    t_.close()
                if gss[0]["update_launches"] > 0:
                    g_us = 1e3 * gss[0]["update_ms_sum"] / gss[0]["update_launches"]
                    g_alg = K_ * 16.0 * R0 * C0
                    g_ach = g_alg / (g_us * 1e-6) / 1e9
                    pmcg, pmcg_src = committed_profile("pmc_traffic.json")
                    g_tr = None
                    if pmcg:
                        hit = [v for k_, v in pmcg.items() if "lpx_group_fused" in k_]
                        if hit:
                            g_tr = max(hit, key=lambda v: v["hbm_bytes_per_launch"])["hbm_bytes_per_launch"]
                    res["kernel"] = {"kernel": "lpx_group_fused", "bound": "hbm", "achieved": g_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": g_ach / HBM_PEAK_GBS, "traffic": g_tr, "traffic_source": pmcg_src if g_tr else None,
                                     "algorithmic_bytes_per_launch": g_alg, "avg_kernel_us": g_us, "launches": gss[0]["update_launches"],
                                     "shape": f"{K_} node tableaux of {R0}x{C0}, every slot live",
                                     "timing": "HIP start/stop events bound to each dispatch on the library stream, this run"}
            out["bnb_warm"] = res
        # ---- a 0/1 IP small enough to be SOLVED: incumbents appear, the all-reduced bound prunes, pools are rebalanced ----
        if leg("bnb_prune"):
            cs, As, rels, bs = synth.binary_ip(args.bnb_prune_n, args.bnb_prune_m)
            ps = L.LPProblem.from_arrays(0, cs, As, rels, bs)
            out["bnb_prune"] = bnb_leg(ps, f"random 0/1 IP n={args.bnb_prune_n} m={args.bnb_prune_m} (+{args.bnb_prune_n} bound rows), repaired mode, "
                                           "sharded level search with the depth-first-K pool (128 node LPs per round, each resident in the LDS of one or two CUs), solved to optimality (no node budget)",
                                       bnb_search=1, bnb_dive=1, concurrent_nodes=128, max_nodes=0)
        # ---- the same with node LPs too large for one CU's LDS: the shared bound on a BASELINE-shaped (mid-size) instance ----
        if leg("bnb_prune_mid"):
            cm, Am, relm, bm = synth.binary_ip(args.bnb_mid_n, args.bnb_mid_m, seed=args.bnb_mid_seed)
            pm = L.LPProblem.from_arrays(0, cm, Am, relm, bm)
            Rm, Cm = args.bnb_mid_m + args.bnb_mid_n + 1, 2 * args.bnb_mid_n + args.bnb_mid_m + 1
            res = bnb_leg(pm, f"random 0/1 IP n={args.bnb_mid_n} m={args.bnb_mid_m} (+{args.bnb_mid_n} bound rows; root tableau {Rm}x{Cm} f64 = "
                              f"{Rm * Cm * 8 / 1024:.0f} KB > one CU's 160 KB of LDS), repaired mode, sharded level search with the depth-first-K "
                              f"pool ({args.bnb_mid_concurrent} node LPs per round), every node re-solved from the slack basis as the reference does"
                              + (f"; GLOBAL budget {args.bnb_mid_nodes} nodes" if args.bnb_mid_nodes else ", solved to optimality (no node budget)"),
                          bnb_search=1, bnb_dive=1, concurrent_nodes=args.bnb_mid_concurrent, max_nodes=args.bnb_mid_nodes)
            res["pruned_by_bound"] = res.pop("_pruned")
            res["incumbent_updates"] = res.pop("_incumbents")
            out["bnb_prune_mid"] = res
        for key in ("bnb", "bnb_warm", "bnb_prune"):
            if key in out:
                out[key].pop("_pruned", None)
                out[key].pop("_incumbents", None)
        if leg("knapsack") or leg("cpu"):
            pk, wk, capk = synth.knapsack(100_000)
        if leg("knapsack"):
            progress("knapsack leg (config 5)")
            # ---- config 5: sharded knapsack (all ranks) ---------------------------------------------------
            kp = L.LPProblem(L.Sense.Max, pk.tolist(), [L.Constraint(wk.tolist(), L.Rel.LE, capk)])
            kn = L.BranchAndBoundKnapsack(max_nodes=args.knap_nodes, concurrent_nodes=512, rank=rank, world=world,
                                          allreduce_max=allreduce_max)
            barrier()
            t1 = time.perf_counter()
            rk = kn.Solve(kp)
            barrier()
            tk = time.perf_counter() - t1
            pop_total, tk_max = reduce_sum_max(rk.Nodes, tk)
            rel_total, _ = reduce_sum_max(rk.Aux[0], tk)
            out["knapsack"] = {"workload": f"0/1 knapsack n=100000 (config 5), best-first B&B, pop budget per rank {args.knap_nodes}",
                               "nodes_per_s": pop_total / tk_max, "popped": pop_total, "relaxations": rel_total,
                               "relaxations_per_s": rel_total / tk_max, "wall_s": tk_max, "incumbent": rk.OptimalValue,
                               "launches": int(rk.Stats["launches"]), "device_call_s": rk.Stats["loop_ms"] / 1e3,
                               "device_call_fraction_of_wall": rk.Stats["loop_ms"] / 1e3 / tk,
                               "bound": "host replay of the reference's pop order (sequential by definition: popped/expanded/"
                                        "relaxations must equal the reference's); the device evaluates the 512 best evaluated "
                                        "leaves ahead of the search per launch (3 relaxations per job, ~1000 jobs per launch)"}
            pmc, pmc_src = committed_profile("pmc_traffic.json")
            if pmc:
                full = [(k, v) for k, v in pmc.items() if "knap_expand_w@grid" in k]
                if full:
                    k_, v_ = max(full, key=lambda kv: kv[1]["launches_fetch"])
                    jobs = int(k_.split("@grid")[1]) // 64
                    out["knapsack"]["kernel"] = {"name": "knap_expand_w", "jobs_per_launch": jobs,
                                                 "hbm_bytes_per_bound": v_["hbm_bytes_per_launch"] / (3.0 * jobs),
                                                 "full_scan_bytes_per_bound": 16.0 * 100_000,
                                                 "source": f"{pmc_src} (committed PMC passes, FETCH x2 + WRITE; not a measurement of this run)"}

    if rank == 0 and not args.no_extras:
        progress("roofline legs")
        if leg("roofline"):
            # ---- measured device-to-device copy bandwidth of THIS box (report both peaks) ------
            src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            dst = torch.empty_like(src)
            dst.copy_(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = 10 * 2.0 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            del src, dst
            torch.cuda.empty_cache()
            # ---- roofline of the dominant kernel of THIS workload: lpx_pivot_fused in the real solve ----------
            # profile = 1: eager launches, every update dispatch bracketed by its own HIP start/stop events on the
            # library's stream (hipExtLaunchKernelGGL); the same pivots as the timed solve (same LP, same start).
            alg = 16.0 * R * C
            dt.restore()
            status, pst = dt.primal_run(L.default_opts(False, batch=args.batch, profile=1, max_iter=args.roofline_pivots))
            k_ms = pst["update_ms_sum"] / max(pst["update_launches"], 1)
            ach = alg / (k_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(R, C, primal=True)
            out["roofline"] = {"kernel": pivot_kernel(R, C)[0].split("::")[1], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_source": traffic_src, "shape": [R, C],
                               "algorithmic_bytes_per_launch": alg, "avg_kernel_us": 1e3 * k_ms,
                               "launches": pst["update_launches"],
                               "timing": "HIP start/stop events bound to each dispatch on the library stream, this run",
                               "rocprof_avg_kernel_us_committed": rocprof_kernel_us(R, C, primal=True),
                               "measured_copy_gbs": copy_gbs, "frac_vs_measured_copy": ach / copy_gbs,
                               "whole_loop_us_per_pivot": 1e3 * loop_ms / max(pivots, 1)}
            # ---- north-star shape: raw 4096x8192 tableau (exactly 256 MiB = the Infinity Cache), forced pivots ----
            HR, HC = 4096, 8192
            Th = synth.raw_tableau(HR, HC)
            hd = L.DeviceTableau.from_host(Th)
            rows, cols = synth.forced_pivot_list(HR, HC, 20 + args.headline_pivots)
            hd.forced_pivots(rows[:20], cols[:20], 0.1)          # warm-up
            _, hst = hd.forced_pivots(rows[20:], cols[20:], 0.1, profile=1, batch=100)
            hk_ms = hst["update_ms_sum"] / max(hst["update_launches"], 1)
            halg = 16.0 * HR * HC
            hach = halg / (hk_ms * 1e-3) / 1e9
            hd.upload(Th)
            _, hst2 = hd.forced_pivots(rows[20:], cols[20:], 0.1, batch=100)
            htraffic, htraffic_src = pmc_traffic(HR, HC)
            out["roofline_north_star"] = {"kernel": update_kernel(HR, HC)[0].split("::")[1], "shape": [HR, HC], "bound": "hbm",
                                          "achieved": hach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": hach / HBM_PEAK_GBS, "traffic": htraffic, "traffic_source": htraffic_src,
                                          "measured_copy_gbs": copy_gbs, "frac_vs_measured_copy": hach / copy_gbs,
                                          "avg_kernel_us": 1e3 * hk_ms, "rocprof_avg_kernel_us_committed": rocprof_kernel_us(HR, HC),
                                          "launches": hst["update_launches"], "algorithmic_bytes_per_launch": halg,
                                          "pivots_per_s_whole_loop": hst2["pivots"] / (hst2["loop_ms"] * 1e-3),
                                          "note": "4096*8192*8 B = 268435456 B is exactly the 256 MiB Infinity Cache: part of this "
                                                  "rate is cache residency; `roofline` above is the pure HBM stream"}
            hd.close()
            del Th
        if leg("config2") or leg("cpu"):
            c2, A2, b2 = synth.dense_lp(1024, 2048)
            T2, basis2 = synth.primal_tableau_from(c2, A2, b2)
        if leg("config2"):
            # ---- config 2 (m=1024 n=2048, 25 MB): resident in LDS -- a latency-bound kernel, no HBM roofline -----
            progress("config 2 leg")
            d2 = L.DeviceTableau.from_host(T2, basis2)
            d2.snapshot()
            d2.primal_run(L.default_opts(False))
            d2.restore()
            t2 = time.perf_counter()
            status, rst = d2.primal_run(L.default_opts(False))
            r_wall = time.perf_counter() - t2
            d2.restore()
            status, sst = d2.primal_run(L.default_opts(False, batch=args.batch, resident=-1))
            d2.restore()
            status, s2p = d2.primal_run(L.default_opts(False, batch=args.batch, resident=-1, profile=1, max_iter=600))
            out["config2"] = {"workload": "dense random LP m=1024 n=2048 (config 2), tableau 1025x3073 f64 = 25 MB, solved to OPTIMAL",
                              "resident": {"kernel": "lpx_resident_primal", "bound": "latency (two cross-CU exchanges per pivot; "
                                                     "tableau in LDS, HBM sees it once per launch)",
                                           "pivots": rst["pivots"], "launches": rst["launches"],
                                           "pivots_per_s": rst["pivots"] / r_wall,
                                           "us_per_pivot": 1e6 * r_wall / max(rst["pivots"], 1)},
                              "streaming": {"kernels": "lpx_pivot_fused_c (one launch per pivot, two 25 MB buffers)", "bound": "Infinity Cache (25 MB tableau)",
                                            "pivots_per_s": sst["pivots"] / (sst["loop_ms"] * 1e-3),
                                            "kernel_avg_us": 1e3 * s2p["update_ms_sum"] / max(s2p["update_launches"], 1)}}
            d2.close()
        if leg("revised"):
            progress("revised leg (config 3)")
            # ---- config 3: revised simplex m=4096 n=8192 -----------------------------------------------------
            c3, A3, b3 = synth.dense_lp(4096, 8192)
            rv = L.DeviceRevised(A3, -c3, b3)
            rv.run(max_iter=20, batch=20)
            rv.close()
            rv = L.DeviceRevised(A3, -c3, b3)
            st3, s3 = rv.run(max_iter=args.revised_iters, batch=50)
            rho3 = rv.residual()
            # per-kernel durations (HIP events bound to each dispatch, eager launches) over the NEXT 200 real iterations: SURVEY 8d asks
            # for per-kernel achieved GB/s, never a fused total against the unfused byte count
            kp = rv.profile(200)
            pmc3, pmc3_src = committed_profile("pmc_traffic.json")

            def rv_kernel(name, us, alg_bytes):
                tr = None
                if pmc3:
                    hit = [v for k_, v in pmc3.items() if k_.startswith(f"lpx::{name}<") or k_.startswith(f"void lpx::{name}<") or f"::{name}<" in k_]
                    if hit:
                        tr = max(hit, key=lambda v: v.get("launches_fetch", 0))["hbm_bytes_per_launch"]
                ach = alg_bytes / (us * 1e-6) / 1e9 if us > 0 else None
                return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": tr, "traffic_source": pmc3_src if tr else None,
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_kernel_us": us, "launches": kp["iterations"],
                        "timing": "HIP start/stop events bound to each dispatch on the library stream, this run"}
            rv.set_refactor_mode(1)
            rv.refactor()                                   # fast form once for its allocations
            rv.run(max_iter=args.revised_iters, batch=50)   # drift the inverse again (same iterations: the count restarts)
            t3 = time.perf_counter()
            rv.refactor()                                   # K7' fast: Newton-Schulz, two 4096^3 contractions on the FP64 matrix cores
            refac_fast_s = time.perf_counter() - t3
            fst = rv.refactor_stats()
            rho3b = rv.residual()
            rv.set_refactor_mode(0)
            t3 = time.perf_counter()
            rv.refactor()                                   # K7' exact: device Gauss-Jordan of the 4096x4096 basis, bit-faithful to Invert
            refac_s = time.perf_counter() - t3
            gemm_tf = 2.0 * 4096 ** 3 * fst["gemm_calls"] / (fst["gemm_ms"] * 1e-3) / 1e12 if fst["gemm_ms"] > 0 else None
            out["revised"] = {"workload": "dense random LP m=4096 n=8192, revised simplex (config 3), "
                                          f"first {s3['pivots']} iterations from the slack basis",
                              "iterations_per_s": s3["pivots"] / (s3["loop_ms"] * 1e-3),
                              "us_per_iteration": 1e3 * s3["loop_ms"] / max(s3["pivots"], 1),
                              "unfused_reference_bytes_per_iteration": 8.0 * (5 * 4096 ** 2 + 4096 * 8192),
                              "engine_bytes_per_iteration": 8.0 * 4096 * 8192 + 16.0 * 4097 * 4097,
                              "engine_dataflow": "rv_price (A^T read once, nt) + rv_pick + rv_upd_ftran (W read+written once: the previous "
                                                 "pivot's rank-1 update fused with d = B^-1 a_q) + rv_select2; 4 launches per iteration",
                              "kernels": {
                                  "rv_price": rv_kernel("rv_price", kp["rv_price"], 8.0 * 4096 * 8192),
                                  "rv_upd_ftran": dict(rv_kernel("rv_upd_ftran", kp["rv_upd_ftran"], 16.0 * 4096 * 4096),
                                                       note="W (134 MB) is at home in the Infinity Cache: part of this rate is cache residency"),
                                  "rv_pick": {"bound": "latency (one workgroup: reduces the 2048 pricing candidates; rv_upd_ftran reads the entering column in place)",
                                              "avg_kernel_us": kp["rv_pick"]},
                                  "rv_select2": {"bound": "latency (one workgroup: ratio test with the 1e-12 hysteresis, pivot row, (pi, z) row, bookkeeping)",
                                                 "avg_kernel_us": kp["rv_select2"]},
                                  "sum_us": kp["rv_price"] + kp["rv_pick"] + kp["rv_upd_ftran"] + kp["rv_select2"]},
                              "drift": {"policy": "residual check every 256 iterations, refactor above 1e-9 (default)",
                                        "residual_after_run": rho3[0], "residual_after_fast_refactor": rho3b[0]},
                              "refactor_exact_s": refac_s,
                              "refactor_exact_algorithmic_gbs": 32.0 * 4096 ** 3 / refac_s / 1e9,
                              "refactor_fast_s": refac_fast_s,
                              "refactor_fast": {"kernel": "dgemm_mfma_f64 (v_mfma_f64_16x16x4_f64)", "bound": "mfma", "achieved": gemm_tf,
                                                "peak": 78.6, "unit": "TFLOP/s", "frac": gemm_tf / 78.6 if gemm_tf else None,
                                                "gemm_calls": fst["gemm_calls"], "gemm_ms": fst["gemm_ms"],
                                                "flops_per_call": 2.0 * 4096 ** 3, "newton_schulz_steps": fst["fast_steps"],
                                                "peak_source": "MI355X FP64 matrix = vector peak 78.6 TFLOP/s (SURVEY 8d; 32 flop/clk/SIMD)",
                                                "measured_mfma_issue_rate_tflops": 36.2,
                                                "measured_mfma_issue_rate_source": "profiles/r02_kbench_mfma_f64_rate.txt: v_mfma_f64_16x16x4_f64 back to back, "
                                                                                   "16 independent accumulators, one wave per SIMD, every CU (tools/kbench/mfma_f64_rate.hip) "
                                                                                   "-- a committed microbenchmark, not a measurement of this run",
                                                "frac_vs_measured_issue_rate": gemm_tf / 36.2 if gemm_tf else None},
                              "refactor_note": "exact = the reference's Invert (4096 Gauss-Jordan steps x 16*m*2m bytes), which the reference runs "
                                               "EVERY iteration; fast = one Newton-Schulz step from the maintained inverse; the engine runs either on demand"}
            rv.close()
            del A3
        if leg("cpu"):
            progress("CPU baselines")
            # ---- CPU baseline: oracle (C port of the reference loops), 1 core, bounded sample of the SAME LP ------
            if world == 1:
                from oracle import oracle as O
                npv = args.cpu_sample_pivots
                Tc, bc = T.copy(), basis.copy()
                t2 = time.perf_counter()
                st_c, tr_c = O.primal_tableau(Tc, bc, max_iter=npv)
                cpu_s = time.perf_counter() - t2
                out["cpu_baseline"] = {"value": len(tr_c) / cpu_s, "unit": "pivots/s", "cores": 1, "kind": "port",
                                       "sample": f"first {len(tr_c)} pivots of the same {R}x{C} LP, oracle/primal.c "
                                                 f"(gcc -O2 -ffp-contract=off, scalar), {cpu_s:.1f} s"}
                progress(f"CPU 1-core done ({cpu_s:.1f} s); all-cores baseline")
                try:
                    model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
                except Exception:
                    model = "unknown"
                out["cpu_baseline"]["cpu_model"] = model
                out["cpu_baseline"]["nproc"] = os.cpu_count()
                out["cpu_baseline"]["affinity_cpus"] = len(os.sched_getaffinity(0))
                # courtesy strong baseline: the same loop with Pivot's rows spread over the host cores of this GPU's share
                # (the box grants 16 CPUs per GPU whatever the affinity mask says; oversubscribed OpenMP teams spin)
                ncores = max(1, min(len(os.sched_getaffinity(0)), 16))
                Tm, bm = T.copy(), basis.copy()
                t2 = time.perf_counter()
                st_m, tr_m = O.primal_tableau(Tm, bm, max_iter=npv, threads=ncores)
                mt_s = time.perf_counter() - t2
                assert np.array_equal(tr_m, tr_c) and np.array_equal(Tm, Tc)
                del Tm, Tc
                out["cpu_baseline"]["all_cores"] = {"value": len(tr_m) / mt_s, "unit": "pivots/s", "cores": ncores,
                                                    "sample": f"same {len(tr_m)} pivots, oracle/primal_mt.c (OpenMP over "
                                                              f"the rows of Pivot, bit-identical), {mt_s:.1f} s"}
                progress(f"all-cores done ({mt_s:.1f} s); config 2 / revised CPU samples")
                Tc2, bc2 = T2.copy(), basis2.copy()
                t2 = time.perf_counter()
                st_c2, tr_c2 = O.primal_tableau(Tc2, bc2, max_iter=1500)
                c2_s = time.perf_counter() - t2
                out["cpu_baseline"]["config2_pivots_per_s"] = len(tr_c2) / c2_s
                out["cpu_baseline"]["config2_sample"] = f"first {len(tr_c2)} pivots of the 1025x3073 LP, 1 core, {c2_s:.1f} s"
                # reference-faithful revised path (Invert every iteration), bounded: 3 iterations at m=1024
                t2 = time.perf_counter()
                rr_c = O.revised_solve(O.Problem(O.MAX, c2, A2, np.zeros(1024, np.int32), b2), max_iter=3)
                cr = time.perf_counter() - t2
                out["cpu_baseline"]["revised_iterations_per_s_m1024"] = len(rr_c.trace) / cr
                out["cpu_baseline"]["revised_sample"] = (f"first {len(rr_c.trace)} iterations at m=1024 n=2048, oracle/revised.c "
                                                         f"(full Invert per iteration as the reference), {cr:.1f} s; "
                                                         "config 3 (m=4096) costs 64x the flops per iteration -- extrapolation, not measured")
                progress("knapsack / B&B CPU samples")
                # bounded CPU samples of the other legs, for the record
                t2 = time.perf_counter()
                rk_c = O.knapsack_solve(O.Problem(O.MAX, pk, wk.reshape(1, -1), [O.LE], [capk]), max_nodes=4000)
                ck = time.perf_counter() - t2
                out["cpu_baseline"]["knapsack_nodes_per_s"] = rk_c.nodes_popped / ck
                out["cpu_baseline"]["knapsack_sample"] = f"first {rk_c.nodes_popped} pops of config 5, oracle/knapsack.c, {ck:.1f} s"
                t2 = time.perf_counter()
                rb_c = O.bnb_solve(O.Problem(O.MAX, cb, Ab, relb.astype(np.int32), bb), 1, max_nodes=5)
                cbn = time.perf_counter() - t2
                out["cpu_baseline"]["bnb_nodes_per_s"] = rb_c.lp_solves / cbn
                out["cpu_baseline"]["bnb_sample"] = f"first {rb_c.lp_solves} LP relaxations of config 4 (repaired, DFS), oracle/bnb.c, {cbn:.1f} s"
    dt.close()
    barrier()
    if engine_comm == "lpx_comm":
        out["config"]["lpx_comm"] = L.comm.info()
        L.comm.destroy()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
