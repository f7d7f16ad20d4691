#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into profiles/<tag>_rocprof_summary.json + .md:
per-kernel time from --kernel-trace --stats, PMC counters per launch of the merge kernel, and the HBM
traffic corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and on gfx950
reads exactly 1/2 of a wide coalesced streaming read's bytes -> doubled; WRITE_SIZE in KiB, exact).
usage: summarize_profile.py <prof dir> <tag> [kernel substring] [algorithmic bytes] [workload] [commit]
With a workload name it also records the measured HBM bytes (and VALU wave-instructions) per launch in profiles/r04_pmc_counters.json - what bench.py
prints as roofline.traffic / roofline_valu - with the commit and the hash of the kernel's source files the counters were collected on.
DATA-DEPENDENT KERNELS: pass the dominant kernel's name substring; counters are averaged over its launches only.
A workload that is several kernels per step (config 3: streaming merge + hot-pixel pass) passes "nameA+nameB": the step time is the
sum of the kernels' average durations and the counters are summed over them (each averaged over its own launches)."""
import csv
import glob
import json
import os
import statistics
import sys

src, tag = sys.argv[1], sys.argv[2]
kname = sys.argv[3] if len(sys.argv) > 3 else "merge_u8_val3"
alg = int(sys.argv[4]) if len(sys.argv) > 4 else 754974720
csv.field_size_limit(1 << 30)
out = {"source": src}
ks = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")
rows = list(csv.DictReader(open(ks[0])))
out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                        "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3, "pct": float(r["Percentage"])}
                       for r in rows[:6]]
knames = kname.split("+")
parts = [next(r for r in out["kernel_stats"] if kn in r["name"]) for kn in knames]
merge = dict(parts[0])
if len(parts) > 1:
    merge["name"] = " + ".join(p_["name"] for p_ in parts)
    for key in ("avg_us", "min_us", "max_us", "pct"):
        merge[key] = sum(p_[key] for p_ in parts)
    out["step_kernels"] = parts
counters = {}
for f in glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"):
    per = {}
    info = {}
    for r in csv.DictReader(open(f)):
        kn = next((k_ for k_ in knames if k_ in r["Kernel_Name"]), None)
        if kn is None:
            continue
        info.setdefault(kn, r)
        per.setdefault((kn, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    tot = {}
    for (kn, k), v in per.items():
        tot[k] = tot.get(k, 0.0) + statistics.mean(v)
    counters.update(tot)
    for kn, r in info.items():
        out.setdefault("launch", {})[kn] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                             "Workgroup_Size", "Grid_Size") if k in r}
out["counters_per_launch"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    rd = counters["FETCH_SIZE"] * 1024 * 2          # gfx950 correction: counter reads 1/2 of wide streaming reads
    wr = counters["WRITE_SIZE"] * 1024
    out["traffic"] = {"fetch_size_kib_raw": counters["FETCH_SIZE"], "write_size_kib_raw": counters["WRITE_SIZE"],
                      "read_bytes_corrected": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                      "algorithmic_bytes_per_launch": alg, "ratio_to_algorithmic": (rd + wr) / alg,
                      "note": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact for 16-B/lane stores"}
c = counters
d = {}
if "SQ_WAVE_CYCLES" in c:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
              "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_FLAT"):
        if k in c:
            d[k + "/WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
    d["lds_bank_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
if "GRBM_GUI_ACTIVE" in c:
    d["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8 / (merge["avg_us"] * 1e3)
if "TCC_HIT_sum" in c:
    d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
out["derived"] = d
out["merge_kernel"] = merge
out["roofline"] = {"achieved_GBps": alg / merge["avg_us"] / 1e3, "frac_of_8TBps": alg / merge["avg_us"] / 1e3 / 8000}
json.dump(out, open(f"profiles/{tag}_rocprof_summary.json", "w"), indent=1)
with open(f"profiles/{tag}_rocprof_summary.md", "w") as f:
    f.write(f"# rocprofv3 summary ({tag})\n\nCommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline"
            + (f" --workload {sys.argv[5]}" if len(sys.argv) > 5 and sys.argv[5] != "cfg2" else "") + "` "
            "and one `--pmc` pass per counter group (tools/profile.sh).\n\n## Kernel stats\n\n| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in out["kernel_stats"]:
        f.write(f"| `{r['name']}` | {r['calls']} | {r['avg_us']:.1f} | {r['min_us']:.1f} | {r['max_us']:.1f} | {r['pct']:.1f} |\n")
    f.write(f"\n{kname}: {alg} algorithmic bytes / {merge['avg_us']:.1f} us = {out['roofline']['achieved_GBps']:.0f} GB/s = "
            f"{out['roofline']['frac_of_8TBps']:.3f} of 8 TB/s\n\n## Launch\n\n{json.dumps(out.get('launch', {}))}\n\n## Counters (mean per launch)\n\n")
    for k, v in sorted(counters.items()):
        f.write(f"- {k}: {v:.6g}\n")
    if "traffic" in out:
        t = out["traffic"]
        f.write(f"\n## HBM traffic\n\nread {t['read_bytes_corrected'] / 1e6:.1f} MB (FETCH_SIZE x 1024 x 2) + write {t['write_bytes'] / 1e6:.1f} MB "
                f"= {t['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch = {t['ratio_to_algorithmic']:.3f} x algorithmic ({alg / 1e6:.1f} MB)\n")
    if d:
        f.write("\n## Derived\n\n" + "\n".join(f"- {k}: {v:.4g}" for k, v in d.items()) + "\n")
if len(sys.argv) > 5:
    # profiles/r04_pmc_counters.json: what bench.py prints as roofline.traffic / roofline_valu - keyed by workload, stamped with the
    # commit AND the hash of the kernel's source files (bench.kernel_source_hash), so a record goes stale with the first edit of its kernel
    import pathlib
    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
    import bench
    tp = pathlib.Path(bench.COUNTER_RECORDS)
    rec = json.load(open(tp)) if tp.exists() else {}
    ent = {"kernel": merge["name"], "avg_us_under_profiler": merge["avg_us"], "commit": sys.argv[6] if len(sys.argv) > 6 else "?",
           "source_hash": bench.kernel_source_hash(sys.argv[5]), "source": f"profiles/{tag}_rocprof_summary.json"}
    if "traffic" in out:
        ent.update(hbm_bytes_per_launch=out["traffic"]["hbm_bytes_per_launch"], ratio_to_algorithmic=out["traffic"]["ratio_to_algorithmic"])
    if "SQ_INSTS_VALU" in counters:
        ent.update(valu_wave_instructions_per_launch=counters["SQ_INSTS_VALU"],
                   valu_busy_frac=(counters.get("SQ_ACTIVE_INST_VALU", 0) / counters["SQ_WAVE_CYCLES"]) if counters.get("SQ_WAVE_CYCLES") else None)
    rec[sys.argv[5]] = ent
    json.dump(rec, open(tp, "w"), indent=1)
print(json.dumps({k: out[k] for k in ("merge_kernel", "launch", "counters_per_launch", "traffic", "derived", "roofline") if k in out}, indent=1))
