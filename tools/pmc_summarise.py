"""Summarises rocprofv3 --pmc counter CSVs per kernel and grid size into profiles/*.json.
Usage: pmc_summarise.py <fetch_dir> <write_dir> <out.json>
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half of a wide coalesced
streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores; both are in KiB."""
import csv, glob, json, sys, collections

def load(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            agg[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for key in sorted(set(fetch) | set(write)):
    name, grid = key
    if not any(k in name for k in ("lpx_update", "lpx_pivot_fused", "lpx_group_fused", "lpx_resident", "rv_price", "rv_upd_ftran", "rv_flush", "knap_expand")):
        continue
    f = fetch.get(key, []); w = write.get(key, [])
    # drop early-exit launches (tail of a batch after the loop has finished: they read the state record and leave)
    def live(v):
        if not v: return v
        med = sorted(v)[len(v) // 2]
        return [x for x in v if x > 16 and x >= 0.35 * med]
    f = live(f); w = live(w)
    if not f or not w:
        continue
    fb = 2.0 * 1024.0 * sum(f) / len(f)        # doubled: gfx950 FETCH_SIZE under-count
    wb = 1024.0 * sum(w) / len(w)
    out[f"{name}@grid{grid}"] = {"launches_fetch": len(f), "launches_write": len(w),
                                 "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb,
                                 "hbm_bytes_per_launch": fb + wb,
                                 "raw_FETCH_SIZE_KiB_mean": sum(f) / len(f), "raw_WRITE_SIZE_KiB_mean": sum(w) / len(w)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
