#!/usr/bin/env python3
"""Timeline of a kernel trace (rocprofv3 --kernel-trace --output-format csv): device busy time, idle gaps and the share of
every kernel inside the LAST burst of work (bursts are separated by idle gaps longer than --split ms).  Used on the warm B&B
leg: `rocprofv3 --kernel-trace --output-format csv -d gpurun_out/warm_tl -- python3 bench.py --only bnb_warm --steps 1 --warmup 0`.
"""
import argparse, csv, glob, collections, sys

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--split", type=float, default=20.0, help="idle gap (ms) that separates bursts")
ap.add_argument("--burst", type=int, default=-1, help="which burst to report (default: the longest)")
a = ap.parse_args()
files = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)
if not files: sys.exit("no kernel_trace.csv under " + a.dir)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
rows.sort()
bursts, cur, end = [], [], None
for r in rows:
    if end is not None and r[0] - end > a.split * 1e6: bursts.append(cur); cur = []
    cur.append(r); end = max(end or 0, r[1])
bursts.append(cur)
print(f"{len(rows)} dispatches, {len(bursts)} bursts: " + ", ".join(f"{(max(x[1] for x in b) - b[0][0]) / 1e6:.1f} ms/{len(b)}" for b in bursts))
b = bursts[a.burst] if a.burst >= 0 else max(bursts, key=lambda b: max(x[1] for x in b) - b[0][0])
t0, t1 = b[0][0], max(x[1] for x in b)
# union of busy intervals, and time with >= 2 kernels in flight
ev = []
for s, e, _, _ in b: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = over = 0; depth = 0; last = t0; gaps = []
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    if depth == 0 and t - last > 0: gaps.append(t - last)
    depth += d; last = t
wall = t1 - t0
print(f"burst wall {wall / 1e6:.2f} ms, device busy {busy / 1e6:.2f} ms ({busy / wall:.3f}), two or more kernels in flight {over / 1e6:.2f} ms ({over / wall:.3f})")
big = [g for g in gaps if g > 20e3]
print(f"idle gaps: {len(gaps)} totalling {sum(gaps) / 1e6:.2f} ms; > 20 us: {len(big)} totalling {sum(big) / 1e6:.2f} ms; > 200 us: {sum(1 for g in gaps if g > 200e3)} totalling {sum(g for g in gaps if g > 200e3) / 1e6:.2f} ms")
# the largest idle gaps with the dispatches around them
order = sorted(b, key=lambda r: r[0])
ends = []; cur_end = t0; biggest = []
for i, (s_, e_, n_, g_) in enumerate(order):
    if s_ > cur_end and i > 0: biggest.append((s_ - cur_end, cur_end - t0, last_name, n_, g_))
    if e_ >= cur_end: cur_end = e_; last_name = f"{n_} grid {g_}"
    elif i == 0: last_name = f"{n_} grid {g_}"
for gap, at, before, after, g_ in sorted(biggest, reverse=True)[:24]:
    print(f"  gap {gap / 1e3:8.1f} us at +{at / 1e6:7.2f} ms  after [{before}]  before [{after} grid {g_}]")
per = collections.defaultdict(lambda: [0, 0])
for s, e, n, g in b: per[(n, g)][0] += 1; per[(n, g)][1] += e - s
for (n, g), (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {t / 1e6:9.2f} ms  {c:7d} x {t / c / 1e3:8.1f} us  {n} grid {g}")
# duration histogram of the busiest kernel (20 us bins) and the idle time right before its launches
top = max(per.items(), key=lambda kv: kv[1][1])[0]
hist = collections.Counter(); tsum = collections.Counter()
for s, e, n, g in b:
    if (n, g) == top: hist[(e - s) // 20000] += 1; tsum[(e - s) // 20000] += e - s
print("duration histogram of", top[0])
for k in sorted(hist): print(f"  {20 * k:4d}-{20 * k + 20:4d} us: {hist[k]:6d} launches, {tsum[k] / 1e6:8.2f} ms")
