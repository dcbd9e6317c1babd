"""Regenerates the figures DESIGN.md / profiles/README.md quote from the COMMITTED artefacts of a round, so that the text cannot drift
from the trace:   python tools/design_figures.py r03 > profiles/r03_figures.md
Sources: profiles/<round>_kernel_by_shape.json (rocprofv3 --kernel-trace of `python bench.py`, aggregated per kernel and grid by
tools/trace_by_shape.py), profiles/<round>_pmc_traffic.json (FETCH x2 + WRITE per launch, tools/pmc_summarise.py) and
profiles/<round>_bench.json (the JSON line of the plain run of the same call)."""
import json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda n: os.path.join(ROOT, "profiles", f"{rnd}_{n}")
ks = json.load(open(P("kernel_by_shape.json")))
pmc = json.load(open(P("pmc_traffic.json")))
bench = json.loads([l for l in open(P("bench.json")) if l.startswith("{")][-1])
PEAK = 8000.0


def k(prefix, grid=None):
    """the (kernel, grid) group with the most live calls whose name contains `prefix` (and whose key contains `grid`)"""
    # a trailing "<" in `prefix` = the kernel's name ends there (lpx_resident_group< does not match lpx_resident_group_r)
    exact = prefix.endswith("<")
    pre = prefix.rstrip("<")
    hit = [(n, v) for n, v in ks.items() if (n.split("@")[0].endswith(pre) if exact else pre in n.split("@")[0]) and (grid is None or f"@grid{grid}" in n)]
    return max(hit, key=lambda nv: nv[1]["live_calls"]) if hit else (None, None)


def traffic(prefix):
    hit = [(n, v) for n, v in pmc.items() if prefix in n.split("@")[0]]
    return max(hit, key=lambda nv: nv[1]["hbm_bytes_per_launch"]) if hit else (None, None)


rows = []


def row(label, prefix, alg_bytes=None, grid=None, pmc_prefix=None, note="-"):
    n, v = k(prefix, grid)
    if not v:
        rows.append(f"| {label} | not in the trace | | | | |")
        return
    us = v["mean_ns_live"] / 1e3
    line = f"| {label} | `{n}` | {v['live_calls']} | {us:.2f} | "
    if alg_bytes:
        gbs = alg_bytes / us / 1e3
        line += f"{alg_bytes / 1e6:.1f} MB -> {gbs / 1e3:.2f} TB/s = **{gbs / PEAK:.3f}** | "
    else:
        line += f"{note} | "
    tn, tv = traffic(pmc_prefix or prefix)
    if tv and alg_bytes:
        line += f"{tv['hbm_bytes_per_launch'] / 1e6:.1f} MB (`{tn}`) = {tv['hbm_bytes_per_launch'] / alg_bytes:.3f}x |"
    else:
        line += "- |"
    rows.append(line)


R, C = 4097, 12289
row("K4f primal step, headline LP 4097x12289", "lpx_pivot_fused", 16.0 * R * C, grid=8488448)
row("K4 in-place update, north-star 4096x8192", "lpx_update_mb", 16.0 * 4096 * 8192, grid=2097152)
row("K4f primal step, config 2 (25 MB, cache resident)", "lpx_pivot_fused_c", 16.0 * 1025 * 3073)
row("K0 resident primal loop, config 2 (one launch per solve)", "lpx_resident_primal", note="latency bound: two cross-CU exchanges per pivot, tableau in LDS")
row("K0r register + LDS resident group, config 4 (twelve nodes on chip; a launch is a whole group: grid = lanes x workgroups per node, nodes of the group)", "lpx_resident_group_r<512, 26", note="latency bound: a launch lasts (nodes of the group / 12 slots) x pivots per node x ~11.2 us")
row("K0r narrow form, mid-size IP nodes (3 workgroups per node, 85 on chip; a launch is a whole group)", "lpx_resident_group_r<512, 40", note="latency bound")
row("K0b LDS-resident group (root LPs, small nodes)", "lpx_resident_group<", note="latency bound")
row("K4g group step, warm config 4 (64-slot grid; live slots vary)", "lpx_group_fused", grid="12058624x1", note="two batches' windows overlap on the device here: durations are not additive -- the kernel by itself is the line under the table")
row("K5 rv_price, config 3", "rv_price", 8.0 * 4096 * 8192)
row("K6+K7 rv_upd_ftran, config 3", "rv_upd_ftran", 16.0 * 4096 * 4096)
row("rv_pick", "rv_pick", note="latency bound (one workgroup)")
row("rv_select2", "rv_select2", note="latency bound (one workgroup)")
row("K7' fast: dgemm_mfma_f64 4096^3", "dgemm_mfma_f64", note="MFMA bound: 2*4096^3 flop per call, see the bench line's revised.refactor_fast")
row("K8 knap_expand_w (~1000 jobs)", "knap_expand_w", note="latency bound (chain of dependent loads per job)")
row("lpx_build_children (warm children, one launch per batch)", "lpx_build_children")
row("lpx_park_many", "lpx_park_many")
row("lpx_gather_solution", "lpx_gather_solution")

print(f"# Figures of round {rnd}, regenerated from the committed artefacts (tools/design_figures.py {rnd})\n")
print("| what | kernel@grid (most frequent) | live launches | mean us (rocprofv3) | algorithmic bytes -> rate = fraction of 8 TB/s | PMC traffic per launch |")
print("|---|---|---|---|---|---|")
print("\n".join(rows))
gk = bench.get("bnb_warm", {}).get("kernel")
tn, tv = traffic("lpx_group_fused")
if tv:
    alg = 64 * 16.0 * 769 * 1281
    print(f"\n`lpx_group_fused` at FULL liveness (64 copies of the config-4 root, tools/k4_headline.py): PMC {tv['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch "
          f"against {alg / 1e6:.1f} MB algorithmic (64 x 16 x 769 x 1281) = **{tv['hbm_bytes_per_launch'] / alg:.3f}x**" +
          (f"; HIP events in the bench line: {gk['avg_kernel_us']:.1f} us = {gk['achieved'] / 1e3:.2f} TB/s = **{gk['frac']:.3f}**." if gk else "."))
print("\n## Bench line of the same call (plain run, no profiler)\n")
rf = bench["roofline"]
print(f"* value **{bench['value']:.0f} pivots/s** ({bench['ms_per_step']:.1f} ms per step of {bench['config']['pivots_per_step']:.0f} pivots); `roofline` "
      f"{rf['kernel']} {rf['avg_kernel_us']:.2f} us by HIP events = **{rf['frac']:.3f}** of 8 TB/s ({rf['achieved'] / 1e3:.2f} TB/s; measured copy {rf['measured_copy_gbs'] / 1e3:.2f} TB/s)")
ns = bench["roofline_north_star"]
print(f"* north star 4096x8192: {ns['kernel']} {ns['avg_kernel_us']:.2f} us = **{ns['frac']:.3f}**, whole loop {ns['pivots_per_s_whole_loop']:.0f} pivots/s")
c2 = bench["config2"]
print(f"* config 2: resident {c2['resident']['pivots_per_s']:.0f} pivots/s ({c2['resident']['us_per_pivot']:.2f} us per pivot), streaming {c2['streaming']['pivots_per_s']:.0f} pivots/s ({c2['streaming']['kernel_avg_us']:.2f} us per launch)")
rv = bench["revised"]
print(f"* config 3: {rv['us_per_iteration']:.1f} us per iteration ({rv['iterations_per_s']:.0f} it/s); per kernel (HIP events): " +
      ", ".join(f"{n} {v['avg_kernel_us']:.1f} us" + (f" = {v['frac']:.3f}" if v.get('frac') else "") for n, v in rv.get("kernels", {}).items() if isinstance(v, dict)))
for key in ("bnb", "bnb_warm", "bnb_prune", "bnb_prune_mid"):
    b = bench.get(key)
    if b:
        extra = f", whole-leg fraction **{b['roofline']['frac']:.3f}**" if "roofline" in b and "streaming" not in b else ""
        extra += f", incumbent {b['incumbent']}" if b.get("incumbent") is not None else ""
        extra += f", pruned by bound {b['pruned_by_bound']:.0f}, incumbent updates {b['incumbent_updates']:.0f}" if "pruned_by_bound" in b else ""
        print(f"* {key}: **{b['nodes_per_s']:.0f} nodes/s** ({b['lp_relaxations']:.0f} LPs, {b['pivots']:.0f} pivots, {b['wall_s']:.3f} s{extra})")
        st = b.get("streaming")
        if st:
            print(f"  * the same search on the streaming kernels (two rolling batches of 64 through lpx_group_fused): {st['nodes_per_s']:.0f} nodes/s "
                  f"({st['wall_s']:.3f} s), whole-leg fraction **{st['roofline']['frac']:.3f}** of 8 TB/s")
kn = bench["knapsack"]
print(f"* knapsack: {kn['nodes_per_s'] / 1e6:.2f} M nodes/s ({kn['popped']:.0f} pops, {kn['launches']} launches, device calls {100 * kn['device_call_fraction_of_wall']:.0f} % of the wall time)")
cb = bench["cpu_baseline"]
print(f"* cpu_baseline: {cb['value']:.1f} pivots/s on 1 core ({cb['sample']}); all cores: {cb['all_cores']['value']:.0f} pivots/s on {cb['all_cores']['cores']}; {cb.get('cpu_model')}")
