#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into the tracked profiles/ folder:
kernel-trace stats (csv copy), PMC counters per launch of the dominant kernel, and
HBM traffic per launch with the gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md:
FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read; WRITE_SIZE is exact;
both are in KiB)."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

src = sys.argv[1]                 # e.g. gpurun_out/prof_r1c
tag = sys.argv[2]                 # e.g. r01_p3tet
kernel_substr = sys.argv[3] if len(sys.argv) > 3 else "tabulate_simplex"
workload = sys.argv[4] if len(sys.argv) > 4 else "p3tet"
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 100000
os.makedirs("profiles", exist_ok=True)

def newest(pattern):
    """gpurun MERGES output directories: a re-run leaves the previous run's files (other PIDs) beside the new ones."""
    return sorted(glob.glob(pattern), key=os.path.getmtime, reverse=True)


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_kernel_stats.csv")
counters = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    for f in newest(os.path.join(d, "*", "*_counter_collection.csv"))[:1]:
        agg = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count",
                                          "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
        for k, v in agg.items():
            # median, not mean: element construction launches the generic kernel once on a tiny
            # batch before the timed launches, which would skew a mean
            counters[k] = {"mean_per_launch": statistics.median(v), "launches": len(v), "statistic": "median"}
        if meta:
            counters["_dispatch"] = meta
dur = None
if stats:
    for r in csv.DictReader(open(stats[0])):
        if kernel_substr in r["Name"]:
            dur = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                   "max_ns": float(r["MaxNs"]), "name": r["Name"]}
            break
# per-phase durations from the kernel trace: bench.py now opens with W + K launches from an idle GPU (ms_per_step_cold), then
# the clock ramp, then the timed W + K and the K launches of the HIP-event timing -- the all-launch average above mixes them
phases = None
traces = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
if traces:
    rows = []
    for r in csv.DictReader(open(traces[0])):
        if kernel_substr in r["Kernel_Name"] and int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) >= 64 * 256:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    d = [x[1] for x in rows]
    if len(d) >= 65:
        phases = {"launches": len(d), "first_25_from_idle_avg_ns": statistics.mean(d[:25]),
                  "last_40_steady_avg_ns": statistics.mean(d[-40:]), "last_40_steady_median_ns": statistics.median(d[-40:]),
                  "note": "last 40 = the timed steps + the HIP-event timing launches of bench.py --steps 20"}
summary = {"source": src, "kernel": dur, "kernel_phases": phases, "counters": counters}
# fp64 matrix-pipe utilisation: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its
# 8 XCDs (each counts the cycles the launch was active): busy SIMD-cycles / (SIMDs x kernel cycles)
mfma_busy = None
if "SQ_VALU_MFMA_BUSY_CYCLES" in counters and "GRBM_GUI_ACTIVE" in counters and counters["GRBM_GUI_ACTIVE"]["mean_per_launch"] > 0:
    mfma_busy = counters["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / (1024.0 * counters["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0)
    summary["mfma_busy_frac"] = {"value": mfma_busy, "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)"}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    fetch = counters["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2      # gfx950: FETCH_SIZE counts 64 B per 128-B request
    write = counters["WRITE_SIZE"]["mean_per_launch"] * 1024
    summary["hbm_traffic"] = {"read_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                              "total_bytes_per_launch": fetch + write,
                              "note": "FETCH_SIZE*1024*2 (gfx950 half-count correction) + WRITE_SIZE*1024"}
    with open(f"profiles/traffic_{workload}.json", "w") as f:
        json.dump({"workload": workload, "batch": batch, "hbm_bytes_per_launch": fetch + write,
                   "read": fetch, "write": write, "mfma_busy_frac": mfma_busy, "from": f"profiles/{tag}_summary.json"}, f, indent=1)
with open(f"profiles/{tag}_summary.json", "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
