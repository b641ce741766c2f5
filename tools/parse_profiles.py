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
# per-kernel average from the trace itself (skipping nothing: warmup included, stated in the file)
ktr = defaultdict(list)
meta = {}
for f in find("trace", "*kernel_trace.csv"):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r.get("Kernel_Name", "")
            ktr[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                 "Workgroup_Size", "Grid_Size")}
summary["kernels"] = {k: {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v), **meta[k]}
                      for k, v in ktr.items() if "cm" in k or "k_" in k}
# ---- PMC passes
def pmc(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in find(sub, "*counter_collection.csv"):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"n": len(v), "mean": sum(v) / len(v)} for c, v in d.items()} for k, d in acc.items()
            if "k_" in k}


summary["pmc_fetch"] = pmc("pmc_fetch")
summary["pmc_write"] = pmc("pmc_write")
summary["pmc_sq"] = pmc("pmc_sq")
summary["pmc_mix"] = pmc("pmc_mix")
for name in ("bench_trace.json", "bench_fetch.json"):
    try:
        with open(os.path.join(out_dir, name)) as fh:
            summary[name] = json.loads(fh.read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        summary[name] = f"unreadable: {e}"
with open(os.path.join(out_dir, f"summary_{tag}.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps({k: v for k, v in summary.items() if k in ("kernels", "pmc_fetch", "pmc_write")}, indent=1)[:3000])
