#!/usr/bin/env python3
"""Condense a gpurun_out/prof tree (tools/gpu_round.sh) into profiles/<round>/: the rocprofv3
--kernel-trace --stats table and the HBM PMC counters of the integrator kernel, with the gfx950
corrections of MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of a wide coalesced read ->
doubled; WRITE_SIZE exact for 16-B/lane streaming stores; both are in KiB)."""
import csv
import json
import os
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/round01"
tag = sys.argv[3] if len(sys.argv) > 3 else "v1"
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "stats", "r01_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
summary = {"tag": tag}
rows = list(csv.DictReader(open(os.path.join(src, "stats", "r01_kernel_stats.csv"))))
for r in rows:
    if "qa_integrate" in r["Name"]:
        summary["kernel"] = r["Name"]
        summary["calls"] = int(r["Calls"])
        summary["avg_ms"] = float(r["AverageNs"]) * 1e-6
        summary["min_ms"] = float(r["MinNs"]) * 1e-6
        summary["max_ms"] = float(r["MaxNs"]) * 1e-6
        summary["percentage"] = float(r["Percentage"])
for name in ("fetch", "write"):
    p = os.path.join(src, f"pmc_{name}", "r01_counter_collection.csv")
    if not os.path.exists(p):
        continue
    vals = []
    for r in csv.DictReader(open(p)):
        if "qa_integrate" in r["Kernel_Name"]:
            vals.append(float(r["Counter_Value"]))
            summary["vgpr_count_field"] = int(r["VGPR_Count"])
            summary["sgpr_count_field"] = int(r["SGPR_Count"])
            summary["lds_block_size"] = int(r["LDS_Block_Size"])
            summary["grid_size"] = int(r["Grid_Size"])
            summary["workgroup_size"] = int(r["Workgroup_Size"])
    if vals:
        summary[f"{name.upper()}_SIZE_KiB_per_launch"] = sum(vals) / len(vals)
    with open(os.path.join(dst, f"{tag}_pmc_{name}.csv"), "w") as f:
        for i, line in enumerate(open(p)):
            if i == 0 or "qa_integrate" in line:
                f.write(line)
if "FETCH_SIZE_KiB_per_launch" in summary and "WRITE_SIZE_KiB_per_launch" in summary:
    summary["hbm_traffic_bytes_per_launch"] = (2.0 * summary["FETCH_SIZE_KiB_per_launch"] +
                                               summary["WRITE_SIZE_KiB_per_launch"]) * 1024.0
    summary["traffic_note"] = "2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE, KiB -> bytes; separate --pmc passes"
for log in ("bench.log",):
    lp = os.path.join(os.path.dirname(src.rstrip("/")), log)
    if os.path.exists(lp):
        for line in open(lp):
            if line.startswith("{"):
                summary["bench_line"] = json.loads(line)
                w = summary["bench_line"]["config"]
                summary["frame"] = w["frame"]
                summary["spp"] = w["spp"]
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))
