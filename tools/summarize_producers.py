#!/usr/bin/env python3
"""profiles/<tag>_producers_rocprof.json from a tools/profile_producers.sh directory: per-kernel stats of k_welford /
k_energy_* and k_welford's HBM traffic per launch (FETCH_SIZE in KiB, doubled on gfx950 for wide streaming reads as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; WRITE_SIZE in KiB exact) next to its algorithmic bytes.
usage: summarize_producers.py <prof dir> <tag>"""
import csv
import glob
import json
import statistics
import sys

src, tag = sys.argv[1], sys.argv[2]
csv.field_size_limit(1 << 30)
out = {"source": src}
rows = list(csv.DictReader(open(glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0])))
out["kernel_stats"] = [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                        "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2)}
                       for r in rows if "k_welford" in r["Name"] or "k_energy" in r["Name"]]
E = 4096 * 4096 * 3
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{src}/pmc_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_welford<true>" in r["Kernel_Name"] or "k_welfordILb1" in r["Kernel_Name"]:
                per.setdefault(c, []).append(float(r["Counter_Value"]))
if per:
    # the bench issues K = 8, 16, 32, 32(+icrf) launches with M2; group the launches by their write size is not possible
    # (always 32 B/element), so report the K = 32 launches: the ones with the largest fetch
    f = sorted(per.get("FETCH_SIZE", []))
    w = per.get("WRITE_SIZE", [])
    if f and w:
        top = f[-max(1, len(f) // 4):]
        rd = statistics.mean(top) * 1024 * 2
        wr = statistics.mean(w) * 1024
        out["welford_K32_traffic_per_launch"] = {"read_bytes": rd, "write_bytes": wr, "algorithmic_bytes": E * (32 + 32),
                                                 "ratio": round((rd + wr) / (E * 64), 4)}
json.dump(out, open(f"profiles/{tag}_producers_rocprof.json", "w"), indent=1)
print(json.dumps(out, indent=1))
