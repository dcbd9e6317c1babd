"""Aggregates a rocprofv3 kernel_trace.csv by (kernel, grid size): calls, mean, median, p10/p90 in ns.
Usage: trace_by_shape.py <kernel_trace.csv> <out.json>.  Early-exit launches (loop already finished) are
reported separately: they are the ones shorter than 35 % of the median of their group."""
import csv, json, statistics, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]))
    agg[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    med = v[len(v) // 2]
    live = [x for x in v if x >= 0.35 * med]
    out[f"{name}@grid{grid[0]}x{grid[1]}"] = {
        "calls": len(v), "live_calls": len(live), "mean_ns_live": round(statistics.fmean(live)) if live else 0,
        "median_ns": med, "p10_ns": v[len(v) // 10], "p90_ns": v[(9 * len(v)) // 10], "total_ms": round(sum(v) / 1e6, 3)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, d in list(out.items())[:14]:
    print(f"{k:70s} calls {d['calls']:7d}  live-mean {d['mean_ns_live']:8d} ns  median {d['median_ns']:8d} ns  total {d['total_ms']:9.1f} ms")
