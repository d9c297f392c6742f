#!/usr/bin/env python3
"""Summarise one tools/gpu_round.sh session (gpurun_out/<tag>/) into profiles/<tag>_*.

  python tools/pmc_summary.py gpurun_out/r01a profiles/r01a [kernel-substring]

Reads the rocprofv3 CSVs (kernel trace + the separate --pmc passes), averages every counter over the
dispatches of the dominant kernel, and derives:
  VALU utilisation  = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)   (active lanes per issued VALU instruction)
  VALU issue busy   = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): share of all SIMD issue cycles that issued VALU
  VALU lane-slots   = issue busy x utilisation: share of all lane-slots (1024 SIMDs x 16 lanes x cycles) that executed
  (also: the share of wave-cycles that issued VALU, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, both in quad-cycles)
  HBM bytes/launch  = 2 * FETCH_SIZE KiB (gfx950 correction for wide reads, MI355X_MICROARCH.md §HBM — an UPPER
                      bound for this kernel's narrow reads) + WRITE_SIZE KiB
  L2 hit rate       = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
Writes <out>_summary.md, <out>_summary.json, copies the kernel-stats CSV, and updates profiles/hbm_traffic.json
(superseded) / profiles/pmc_views.json (the `traffic` and `executed` fields of bench.py's roofline)."""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    kern = sys.argv[3] if len(sys.argv) > 3 else "render_kernel"
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    counters = defaultdict(list)
    meta = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*_counter_collection.csv"), recursive=True)):
        per_dispatch = defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if kern not in r["Kernel_Name"]:
                continue
            per_dispatch[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count",
                                      "Accum_VGPR_Count", "SGPR_Count")}
        for d in per_dispatch.values():
            for k, v in d.items():
                counters[k].append(v)
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    stats_rows = []
    for f in glob.glob(os.path.join(src, "prof_trace", "**", "*_kernel_stats.csv"), recursive=True):
        shutil.copy(f, out + "_kernel_stats.csv")
        stats_rows = [r for r in csv.DictReader(open(f))]
    kstat = next((r for r in stats_rows if kern in r["Name"]), None)
    d = {}
    g = avg.get
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        d["valu_utilization_pct"] = 100.0 * g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)
    if g("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_ACTIVE_INST_VALU", "valu_issue_share_of_wave_cycles_pct"), ("SQ_WAIT_ANY", "wait_any_share_pct"),
                        ("SQ_WAIT_INST_ANY", "wait_inst_share_pct")):
            if g(k) is not None:
                d[name] = 100.0 * g(k) / g("SQ_WAVE_CYCLES")
    if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
        # SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE is summed over the 8 XCDs
        d["valu_issue_busy_pct"] = 100.0 * 4.0 * g("SQ_ACTIVE_INST_VALU") / (g("GRBM_GUI_ACTIVE") / 8.0 * 1024.0)
        if "valu_utilization_pct" in d:
            d["valu_lane_slot_pct"] = d["valu_issue_busy_pct"] * d["valu_utilization_pct"] / 100.0
    if g("SQ_BUSY_CYCLES") and g("SQ_WAVE_CYCLES"):
        d["mean_waves_in_flight_per_se"] = g("SQ_WAVE_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        d["valu_insts_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
        d["l2_hit_rate_pct"] = 100.0 * g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        d["hbm_read_bytes_per_launch_upper"] = 2.0 * g("FETCH_SIZE") * 1024.0
        d["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
        d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch_upper"] + d["hbm_write_bytes_per_launch"]
        if kstat:
            d["hbm_gbs_achieved"] = d["hbm_bytes_per_launch"] / (float(kstat["AverageNs"]) * 1e-9) / 1e9
    # which build was profiled: the bench line inside the passes' logs carries the library's build id (rt_describe: "build=<hash of
    # the device sources and flags>"); HEAD is this checkout's commit when the summary is written, "+" when the device sources in
    # the tree are no longer the ones that were profiled
    build = None
    for f in sorted(glob.glob(os.path.join(src, "*.log"))):
        m = re.search(r"build=([0-9a-f]{12})", open(f, errors="replace").read())
        if m:
            build = m.group(1)
            break
    head = None
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        sys.path.insert(0, root)
        from importlib import import_module
        if head and build and import_module("ray-tracing_amd._build").source_id() != build:
            head += "+"
    except Exception:
        pass
    summary = {"source": src, "build": build, "head": head, "kernel": meta.get("Kernel_Name", kern), "dispatch": meta, "kernel_stats": kstat,
               "counters_avg_per_launch": avg, "derived": d}
    json.dump(summary, open(out + "_summary.json", "w"), indent=1)
    with open(out + "_summary.md", "w") as f:
        f.write(f"# rocprofv3 summary — {os.path.basename(out)}\n\nsource: `{src}` (tools/gpu_round.sh), kernel `{meta.get('Kernel_Name', kern)}`\n\n")
        f.write(f"build `{build}` (hash of the device sources and flags, as the library reports it in `rt_describe`), summary written at HEAD `{head}`\n\n")
        f.write("command per pass: `rocprofv3 [--kernel-trace --stats | --pmc <counters>] --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32`\n\n")
        if kstat:
            f.write(f"kernel trace: {kstat['Calls']} launches, average {float(kstat['AverageNs'])/1e6:.3f} ms "
                    f"(min {float(kstat['MinNs'])/1e6:.3f}, max {float(kstat['MaxNs'])/1e6:.3f}), {kstat['Percentage']} % of GPU time\n\n")
        if meta:
            f.write("dispatch: grid {Grid_Size}, workgroup {Workgroup_Size}, VGPR {VGPR_Count} + AGPR {Accum_VGPR_Count}, SGPR {SGPR_Count}, "
                    "LDS {LDS_Block_Size} B, scratch {Scratch_Size} B/lane  (register counts as rocprofv3 reports them: per-lane allocation in its own granule units, see tools/kernel_resources.sh for the compiler's counts)\n\n".format(**meta))
        f.write("| derived | value |\n|---|---|\n")
        for k, v in d.items():
            f.write(f"| {k} | {v:,.3f} |\n")
        f.write("\n| counter (mean per launch) | value |\n|---|---|\n")
        for k in sorted(avg):
            f.write(f"| {k} | {avg[k]:,.0f} |\n")
    print(open(out + "_summary.md").read())
    if len(sys.argv) > 4:  # key "<scene>_<W>x<H>_<spp>spp_<precision>": the hardware view bench.py prints as roofline.executed / .traffic
        p = os.path.join(os.path.dirname(out) or ".", "pmc_views.json")
        rec = json.load(open(p)) if os.path.exists(p) else {}
        view = {"from": os.path.basename(out), "build": build, "head": head}
        if "hbm_bytes_per_launch" in d:
            view["hbm_bytes_per_launch"] = d["hbm_bytes_per_launch"]
        if "valu_utilization_pct" in d:
            view["valu_lanes_active_frac"] = round(d["valu_utilization_pct"] / 100.0, 4)
        if "valu_issue_busy_pct" in d:
            view["valu_issue_busy_frac"] = round(d["valu_issue_busy_pct"] / 100.0, 4)
        if "valu_lane_slot_pct" in d:
            view["valu_lane_slot_frac"] = round(d["valu_lane_slot_pct"] / 100.0, 4)
        rec[sys.argv[4]] = view
        json.dump(rec, open(p, "w"), indent=1)


if __name__ == "__main__":
    main()
