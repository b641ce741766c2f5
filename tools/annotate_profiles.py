"""Condense the rocprofv3 summaries of tools/profile_sides.sh (gpurun_out/prof_<tag>_*/summary_*.json) into the files kept
under profiles/: per workload the dominant kernel with its average duration (kernel trace), the register / scratch / LDS
numbers of the CODE OBJECT (tools/kernel_resources.py -- the VGPR_Count column of the rocprofv3 trace reports allocation
granules, not registers), HBM traffic (FETCH_SIZE x 2 x 32 B... see below), SQ wave-cycle split and the fp64 instruction mix.

    python tools/annotate_profiles.py r02 [libcmad_hip.so]  ->  profiles/<tag>_rocprof_summary.json, profiles/<tag>_side_kernels.txt

Units: FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3); on gfx950 FETCH_SIZE counts half of a wide coalesced read
(/opt/skills/guides/MI355X_MICROARCH.md), so HBM bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024.  SQ_* cycle counters are
quad-cycles summed over waves; fractions are taken against SQ_WAVE_CYCLES.  fp64 VALU peak = 78.6 TFLOP/s (half of the
157.3 TFLOP/s fp32 vector rate of the guide); flops = (2 FMA + MUL + ADD) x 64 lanes x mean active-lane fraction.
"""
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
so = sys.argv[2] if len(sys.argv) > 2 else "cmad_amd/csrc/libcmad_hip.so"
def short(n):
    """kernel name without its argument list"""
    m = re.search(r"\((cm_model_desc|long|double|int|unsigned)", n)
    return (n[:m.start()] if m else n).strip()


res = {}
for r in kr.kernels(so):
    # the library holds two builds of the kernel file with the same kernel names (base first, then the HNN build, cmad_amd/build.py);
    # the bench workloads run the base build: keep the first occurrence
    res.setdefault(short(r[0]), {"vgpr": r[1], "agpr": r[5], "sgpr": r[2], "scratch_bytes": r[3], "lds_bytes": r[4]})

out, lines = {}, []
for f in sorted(glob.glob(f"gpurun_out/prof_{tag}_*/summary_{tag}_*.json")):
    d = json.load(open(f))
    wl = d["tag"][len(tag) + 1:]
    bench = d.get("bench_trace.json", {})
    kernels = {k: v for k, v in d["kernels"].items() if "reduce" not in k and "sum_rows" not in k and "k_screen_reset" not in k
               and "k_pool_ticket_zero" not in k}            # (one-thread helper launches: microseconds, same grid at every batch size)
    # the workload's own kernel: the one with the most time that is not the 65536-point self-check launch
    main = max(kernels, key=lambda k: (kernels[k]["calls"], kernels[k]["avg_us"]))
    kv = kernels[main]
    name = short(main)
    entry = {"kernel": name, "calls": kv["calls"], "avg_us": kv["avg_us"], "min_us": kv["min_us"], "max_us": kv["max_us"],
             "code_object": res.get(name), "workload": bench.get("config", {}).get("workload") if isinstance(bench, dict) else None}
    pts = bench.get("config", {}).get("points_per_gpu") if isinstance(bench, dict) else None
    bpu = bench.get("roofline", {}).get("algorithmic_bytes_per_update") if isinstance(bench, dict) else None
    # a workload whose step is several kernels (work-pool update + tangent / reverse at the stored states): the roofline
    # fraction of the STEP divides the algorithmic bytes by the sum of their average durations, not by the dominant one's
    route = {short(k): v["avg_us"] for k, v in kernels.items() if v["calls"] == kv["calls"]}
    if "k_reverse" in name:          # a fused kernel IS its step (the headline run also times the objective and two side variants)
        route = {name: kv["avg_us"]}
    entry["step_kernels_us"] = route
    step_us = sum(route.values())
    # the same sums over the first 14 launches and over the launches from the 50th on (parse_profiles.py)
    by_short = {short(k): v for k, v in kernels.items()}
    part = {w: sum(by_short[k].get(w, float("nan")) for k in route) for w in ("first14_avg_us", "sustained_avg_us")}
    entry.update({k: v for k, v in part.items() if v == v})
    if pts and bpu:
        entry["algorithmic_GBs_at_kernel_time"] = bpu * pts / (step_us * 1e-6) / 1e9
        entry["hbm_roof_frac_kernel_only"] = entry["algorithmic_GBs_at_kernel_time"] / 8000.0
        for w, key in (("first14_avg_us", "hbm_roof_frac_first14"), ("sustained_avg_us", "hbm_roof_frac_sustained")):
            if part[w] == part[w]:
                entry[key] = bpu * pts / (part[w] * 1e-6) / 1e9 / 8000.0
        entry["bench_under_trace"] = {k: bench.get(k) for k in ("value", "ms_per_step")}
    c = {}
    for sec in ("pmc_sq", "pmc_mix32", "pmc_mix", "pmc_fetch", "pmc_write"):
        for k, v in d.get(sec, {}).items():
            if short(k) == name:
                c.update({cn: cv["mean"] for cn, cv in v.items()})
    if c:
        entry["counters_mean_per_launch"] = c
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            entry["wave_cycle_split"] = {"valu_active": c.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_any": c.get("SQ_WAIT_ANY", 0) / wc,
                                         "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / wc}
        if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
            entry["active_lane_fraction"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            entry["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
            # a step of several kernels (screened update: k_screen + k_update_listed; work pool + reverse): the step's traffic is
            # the sum over its kernels
            step_bytes = 0.0
            for k in kernels:
                if short(k) in route:
                    fe = d.get("pmc_fetch", {}).get(k, {}).get("FETCH_SIZE", {}).get("mean", 0.0)
                    wr = d.get("pmc_write", {}).get(k, {}).get("WRITE_SIZE", {}).get("mean", 0.0)
                    step_bytes += fe * 1024 * 2 + wr * 1024
            entry["step_hbm_bytes_per_launch"] = step_bytes
            if pts:
                entry["hbm_bytes_per_point"] = step_bytes / pts
                entry["dominant_kernel_hbm_bytes_per_point"] = entry["hbm_bytes_per_launch"] / pts
        if "SQ_INSTS_VALU_FMA_F64" in c:
            lanes = entry.get("active_lane_fraction", 1.0)
            flops = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]) * 64.0 * lanes
            entry["fp64_tflops"] = flops / (kv["avg_us"] * 1e-6) / 1e12
            entry["fp64_valu_roof_frac"] = entry["fp64_tflops"] / 78.6
            if c.get("SQ_WAVES"):
                entry["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"]
            # share of the VALU stream that is floating-point arithmetic: fp64 (FMA + MUL + ADD + TRANS) and, where the float pass
            # ran, fp32 (the warm-start seeds)
            if c.get("SQ_INSTS_VALU"):
                f64 = sum(c.get(f"SQ_INSTS_VALU_{k}_F64", 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
                f32 = sum(c.get(f"SQ_INSTS_VALU_{k}_F32", 0.0) for k in ("FMA", "MUL", "ADD", "TRANS"))
                entry["fp64_share_of_valu"] = f64 / c["SQ_INSTS_VALU"]
                if "SQ_INSTS_VALU_FMA_F32" in c:
                    entry["fp32_share_of_valu"] = f32 / c["SQ_INSTS_VALU"]
        if c.get("GRBM_GUI_ACTIVE"):
            # busy cycles summed over the 8 XCDs during the launch -> mean engine clock while the kernel ran (approximate: the
            # counter pass and the kernel-trace pass are different runs)
            entry["approx_engine_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (kv["avg_us"] * 1e3)
    out[wl] = entry
    co = entry["code_object"] or {}
    lines.append(f"{wl:16s} {name[len('void (anonymous namespace)::'):][:52]:52s} avg {kv['avg_us']:8.1f} us x{kv['calls']:3d} | "
                 f"{co.get('vgpr', '?')} vgpr ({co.get('agpr', '?')} acc) {co.get('scratch_bytes', '?')} B scratch {co.get('lds_bytes', '?')} B lds | "
                 f"hbm roof {entry.get('hbm_roof_frac_kernel_only', float('nan')):.3f} (first 14: {entry.get('hbm_roof_frac_first14', float('nan')):.3f}, "
                 f"from launch 50: {entry.get('hbm_roof_frac_sustained', float('nan')):.3f}) | fp64 valu roof {entry.get('fp64_valu_roof_frac', float('nan')):.3f} | "
                 f"valu/wait/stall {entry.get('wave_cycle_split', {}).get('valu_active', float('nan')):.2f}/"
                 f"{entry.get('wave_cycle_split', {}).get('wait_any', float('nan')):.2f}/"
                 f"{entry.get('wave_cycle_split', {}).get('wait_inst_any', float('nan')):.2f} | lanes {entry.get('active_lane_fraction', float('nan')):.2f} | "
                 f"traffic {entry.get('hbm_bytes_per_point', float('nan')):.1f} B/pt | {entry.get('valu_insts_per_wave', float('nan')):.0f} valu/wave "
                 f"(fp64 {entry.get('fp64_share_of_valu', float('nan')):.2f} fp32 {entry.get('fp32_share_of_valu', float('nan')):.2f}) | "
                 f"~{entry.get('approx_engine_clock_GHz', float('nan')):.2f} GHz")
os.makedirs("profiles", exist_ok=True)
json.dump(out, open(f"profiles/{tag}_rocprof_summary.json", "w"), indent=1)
open(f"profiles/{tag}_side_kernels.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
