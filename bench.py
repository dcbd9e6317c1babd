#!/usr/bin/env python3
"""bench.py -- headline benchmark of the lpx simplex hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one complete device-resident solve of BASELINE.json's config 2 (dense random LP
m=1024, n=2048, primal tableau simplex, seed 20251003) starting from the slack basis: the
tableau is restored from a pristine HBM snapshot (D2D, inside the timed region) and the
select/update loop runs to OPTIMAL.  `value` = pivots completed by all ranks / max-over-ranks
wall time.  A single LP does not shard (DESIGN.md "Multi-GPU"): with --gpus N every rank
solves its own replica of the workload on its own GPU ("replicas only", weak scaling).

Extra objects on the same JSON line:
  roofline          rank-1 update kernel on THIS workload: algorithmic bytes (16*R*C per pivot)
                    / average kernel duration measured with HIP events around each launch on
                    the library's stream (profile pass over the same solve, rank 0).
  roofline_headline the same kernel on the north-star shape, raw tableau 4096x8192 FP64
                    (268 MB > 256 MiB Infinity Cache, a true HBM stream), >=200 timed pivots.
  cpu_baseline      the CPU oracle (C restatement of the reference's scalar loops, 1 core) on
                    a bounded sample of the same LP (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "simplex pivots/sec on dense m×n tableau; B&B nodes/sec at 1/2/4/8 GPUs"
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--m", type=int, default=1024)
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline/cpu_baseline legs")
    ap.add_argument("--headline-pivots", type=int, default=200)
    ap.add_argument("--cpu-sample-pivots", type=int, default=10000)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import numpy as np
    import torch
    import torch.distributed as dist

    import linear_programming_solver_lpr381_amd as L
    from linear_programming_solver_lpr381_amd import synth

    torch.cuda.set_device(local_rank)
    L._lib.check(L._lib.lib().lpx_init(local_rank))
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- workload: config 2 -------------------------------------------------------------------
    m, n = args.m, args.n
    c, A, b = synth.dense_lp(m, n, seed=synth.SEED + rank)   # one replica per rank, own seed
    T, basis = synth.primal_tableau_from(c, A, b)
    R, C = T.shape
    dt = L.DeviceTableau.from_host(T, basis)
    dt.snapshot()
    opts = L.default_opts(False, batch=args.batch, use_graph=0 if args.no_graph else 1)

    def step():
        dt.restore()
        status, st = dt.primal_run(opts)
        assert status == 0, f"solve ended with status {status}"
        return st

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

    tot = torch.tensor([float(pivots), dt_s], dtype=torch.float64, device="cuda")
    if world > 1:
        t_max = tot[1:2].clone()
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        p_sum = tot[0:1].clone()
        dist.all_reduce(p_sum, op=dist.ReduceOp.SUM)
        total_pivots, wall = float(p_sum.item()), float(t_max.item())
    else:
        total_pivots, wall = float(pivots), dt_s

    out = {
        "metric": METRIC,
        "value": total_pivots / wall,
        "unit": "pivots/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * wall / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"dense random LP m={m} n={n}, primal tableau simplex (config 2), "
                        f"tableau {R}x{C} f64, solved to OPTIMAL from the slack basis",
            "parallelism": "replicas only (one LP per GPU)" if world > 1 else "1 GPU",
            "pivots_per_step": pivots / max(args.steps, 1),
            "batch": args.batch,
            "hipgraph": not args.no_graph,
            "device_loop_ms_per_step": loop_ms / max(args.steps, 1),
        },
    }

    if rank == 0 and not args.no_extras:
        # ---- roofline of the rank-1 update kernel on this workload (HIP events, profile pass) ----
        popts = L.default_opts(False, batch=args.batch, profile=1)
        dt.restore()
        status, pst = dt.primal_run(popts)
        k_ms = pst["update_ms_sum"] / max(pst["update_launches"], 1)
        alg = 16.0 * R * C
        ach = alg / (k_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "lpx_update", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                           "avg_kernel_us": 1e3 * k_ms, "launches": pst["update_launches"],
                           "algorithmic_bytes_per_launch": alg,
                           "note": "25 MB tableau: resident in the 256 MiB Infinity Cache, not an HBM stream"}
        # ---- headline shape: raw 4096x8192 tableau, forced pivots -----------------------------------
        HR, HC = 4096, 8192
        Th = synth.raw_tableau(HR, HC)
        hd = L.DeviceTableau.from_host(Th)
        rows, cols = synth.forced_pivot_list(HR, HC, 20 + args.headline_pivots)
        hd.forced_pivots(rows[:20], cols[:20], 0.1)          # warm-up
        _, hst = hd.forced_pivots(rows[20:], cols[20:], 0.1, profile=1, batch=100)
        hk_ms = hst["update_ms_sum"] / max(hst["update_launches"], 1)
        halg = 16.0 * HR * HC
        hach = halg / (hk_ms * 1e-3) / 1e9
        # whole-loop pivots/s on the headline shape (graph replay, select + update)
        hd.upload(Th)
        t1 = time.perf_counter()
        _, hst2 = hd.forced_pivots(rows[20:], cols[20:], 0.1, batch=100)
        hwall = time.perf_counter() - t1
        out["roofline_headline"] = {"kernel": "lpx_update", "shape": [HR, HC], "bound": "hbm",
                                    "achieved": hach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": hach / HBM_PEAK_GBS, "traffic": None,
                                    "avg_kernel_us": 1e3 * hk_ms, "launches": hst["update_launches"],
                                    "algorithmic_bytes_per_launch": halg,
                                    "pivots_per_s_whole_loop": hst2["pivots"] / (hst2["loop_ms"] * 1e-3),
                                    "host_wall_s": hwall}
        hd.close()
        # ---- CPU baseline: oracle (C port of the reference loops), 1 core, bounded sample ----------
        if world == 1:
            from oracle import oracle as O
            Tc, bc = T.copy(), basis.copy()
            k = args.cpu_sample_pivots
            t2 = time.perf_counter()
            st_c, tr_c = O.primal_tableau(Tc, bc, max_iter=k)
            cpu_s = time.perf_counter() - t2
            out["cpu_baseline"] = {"value": len(tr_c) / cpu_s, "unit": "pivots/s", "cores": 1, "kind": "port",
                                   "sample": f"first {len(tr_c)} pivots of the same {R}x{C} LP, oracle/primal.c "
                                             f"(gcc -O2 -ffp-contract=off, scalar), {cpu_s:.1f} s"}
    dt.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
