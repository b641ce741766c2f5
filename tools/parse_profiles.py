"""Condense rocprofv3 CSV output (kernel-trace stats + PMC passes) into small summaries.
Usage: python tools/parse_profiles.py <gpurun_out/prof_TAG> <TAG>  -> gpurun_out/prof_TAG/summary_*.{csv,json}"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]


def find(sub, pattern):
    return sorted(glob.glob(os.path.join(out_dir, sub, "**", pattern), recursive=True))


summary = {"tag": tag}
# ---- kernel stats
rows = []
for f in find("trace", "*kernel_stats.csv"):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
with open(os.path.join(out_dir, f"summary_kernel_stats_{tag}.csv"), "w") as fh:
    if rows:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow(r)
# Per-kernel average from the trace itself, grouped by (kernel name, grid size): bench.py's 65 536-point self-check and any
# other launch of the same kernel on a different batch are different grids and never enter the average of the full-size
# launches (round 2 averaged per NAME and understated the update-only kernels by 1/15).  The summary of a kernel is its
# LARGEST grid; the other grids are listed under "other_grids".  Warm-up launches of the same grid are included.
ktr = defaultdict(list)
meta = {}
for f in find("trace", "*kernel_trace.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            key = (r.get("Kernel_Name", ""), int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0))
            ktr[key].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
            meta[key] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size")}
            meta[key]["Workgroup_Size"] = r.get("Workgroup_Size") or r.get("Workgroup_Size_X")
            meta[key]["Grid_Size"] = key[1]


def stats(tv):
    """tv: (start timestamp, duration us) of every launch.  A back-to-back sequence of an fp64-heavy kernel dips in clock over its
    launches ~3-30 and recovers (profiles/r04_sustained.txt): beside the plain average, the first 14 launches and -- for a
    sequence of 100 or more -- the launches from the 50th on are averaged separately."""
    v = [d for _, d in sorted(tv)]
    out = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
    if len(v) >= 14:
        out["first14_avg_us"] = sum(v[:14]) / 14
    if len(v) >= 100:
        out["sustained_avg_us"] = sum(v[50:]) / len(v[50:])
    return out


main_grid = {}
for (name, grid) in ktr:
    main_grid[name] = max(main_grid.get(name, 0), grid)
summary["kernels"] = {}
for (name, grid), v in ktr.items():
    if not ("cm" in name or "k_" in name) or grid != main_grid[name]:
        continue
    others = {str(g): stats(w) for (n2, g), w in ktr.items() if n2 == name and g != grid}
    summary["kernels"][name] = {**stats(v), **meta[(name, grid)], **({"other_grids": others} if others else {})}
# ---- PMC passes
def pmc(sub):
    """counter means over the launches of each kernel's largest grid only (same filter as the kernel trace)"""
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[(r["Kernel_Name"], int(r.get("Grid_Size") or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    top = {}
    for (name, grid) in acc:
        top[name] = max(top.get(name, 0), grid)
    return {name: {c: {"n": len(v), "mean": sum(v) / len(v), "grid": grid} for c, v in d.items()}
            for (name, grid), d in acc.items() if "k_" in name and grid == top[name]}


summary["pmc_fetch"] = pmc("pmc_fetch")
summary["pmc_write"] = pmc("pmc_write")
summary["pmc_sq"] = pmc("pmc_sq")
summary["pmc_mix"] = pmc("pmc_mix")
summary["pmc_mix32"] = pmc("pmc_mix32")
for name in ("bench_trace.json", "bench_fetch.json"):
    try:
        with open(os.path.join(out_dir, name)) as fh:
            summary[name] = json.loads(fh.read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        summary[name] = f"unreadable: {e}"
with open(os.path.join(out_dir, f"summary_{tag}.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: v for k, v in summary.items() if k in ("kernels", "pmc_fetch", "pmc_write")}, indent=1)[:3000])
