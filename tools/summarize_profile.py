#!/usr/bin/env python
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile.sh on the GPU box) into tracked files under profiles/."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
kernel = sys.argv[2] if len(sys.argv) > 2 else "pt_megakernel"
src = ROOT / "gpurun_out" / f"prof_{tag}"
dst = ROOT / "profiles"
dst.mkdir(exist_ok=True)

stats = glob.glob(str(src / "trace" / "*" / "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst / f"{tag}_kernel_stats.csv")


def headline(rows):
    """The launches that count: a handle's first launch is a one-chunk probe of the record density (1 ms), and side measurements
    may use the same kernel on smaller jobs -- keep the dispatches that last at least half as long as the longest one."""
    rows = [r for r in rows if kernel in r["Kernel_Name"]]
    if not rows:
        return []
    dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    longest = max(dur(r) for r in rows)
    return [r for r in rows if dur(r) * 2 >= longest]


pmc = collections.OrderedDict()
for f in sorted(glob.glob(str(src / "pmc_*" / "*" / "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in headline(list(csv.DictReader(open(f)))):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
summary = {"tag": tag, "kernel": kernel, "command": "rocprofv3 --kernel-trace --stats / --pmc <one group per pass> -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary",
           "launches_counted": "dispatches of the kernel lasting at least half as long as the longest one (a handle's first launch is a one-chunk record-density probe)",
           "pmc": pmc}
traces = glob.glob(str(src / "trace" / "*" / "*_kernel_trace.csv"))
if traces:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in headline(list(csv.DictReader(open(traces[0]))))]
    if d:
        summary["kernel_trace"] = {"calls": len(d), "average_ns": sum(d) / len(d), "min_ns": float(min(d)), "max_ns": float(max(d))}
if stats and "kernel_trace" in summary:
    for r in csv.DictReader(open(stats[0])):
        if kernel in r["Name"]:
            summary["kernel_trace"]["stats_csv"] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "percentage": float(r["Percentage"]),
                                                    "note": "rocprofv3 --stats averages ALL dispatches of the kernel, the probe included"}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes read.
    # Calibrated on this kernel's own known read: the framebuffer read-modify-write reads W*H*12 B per launch.
    fetch = pmc["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    write = pmc["WRITE_SIZE"]["mean_per_launch"] * 1024
    summary["hbm"] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "hbm_bytes_per_launch": fetch + write,
                      "correction": "FETCH_SIZE x 2 (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as is"}
    (dst / f"{tag}_hbm_traffic.json").write_text(json.dumps({"hbm_bytes_per_launch": fetch + write, **summary["hbm"]}, indent=2) + "\n")
if {"SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"} <= pmc.keys():
    summary["valu_lane_utilisation"] = pmc["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (pmc["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] * 64)
if "GRBM_GUI_ACTIVE" in pmc and "kernel_trace" in summary:
    summary["effective_clock_ghz"] = pmc["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8 / summary["kernel_trace"]["average_ns"]
if "SQ_INSTS_VALU" in pmc and "effective_clock_ghz" in summary:
    # VALU-issue view of the kernel (bench.py copies this object into roofline.valu): SIMD cycles available per VALU
    # wave-instruction against the measured issue floor of 2.04 cycles (tools/ifetch_rate.hip, profiles/r01_ifetch_rate.txt)
    simd_cycles = summary["kernel_trace"]["average_ns"] * summary["effective_clock_ghz"] * 1024
    cpv = simd_cycles / pmc["SQ_INSTS_VALU"]["mean_per_launch"]
    summary["valu"] = {"valu_wave_instructions": pmc["SQ_INSTS_VALU"]["mean_per_launch"], "cycles_per_valu": round(cpv, 3),
                       "issue_floor_cycles": 2.04, "issue_frac": round(2.04 / cpv, 4),
                       "lane_util": round(summary.get("valu_lane_utilisation", 0.0), 4), "clock_ghz": round(summary["effective_clock_ghz"], 3),
                       "wait_inst_any_frac": round(pmc["SQ_WAIT_INST_ANY"]["mean_per_launch"] / pmc["SQ_WAVE_CYCLES"]["mean_per_launch"], 4) if {"SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"} <= pmc.keys() else None,
                       "wait_any_frac": round(pmc["SQ_WAIT_ANY"]["mean_per_launch"] / pmc["SQ_WAVE_CYCLES"]["mean_per_launch"], 4) if {"SQ_WAIT_ANY", "SQ_WAVE_CYCLES"} <= pmc.keys() else None}
(dst / f"{tag}_summary.json").write_text(json.dumps(summary, indent=2) + "\n")
print(json.dumps({k: v for k, v in summary.items() if k != "pmc"}, indent=2))
