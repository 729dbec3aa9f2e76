"""SQ counter passes (rocprofv3 --pmc, tools/gpu_pmc_lib.sh) -> one JSON with the raw sums and the derived compute-side figures.
   python tools/sq_summary.py gpurun_out/pmc_lib OUT.json scene W H spp
Derived (MI355X_MICROARCH.md: a wave64 vector instruction issues over 2 cycles on a SIMD-32; SQ_BUSY_CYCLES is per shader
engine, 32 of them):
   kernel_cycles        = SQ_BUSY_CYCLES / 32
   valu_issue_frac      = SQ_INSTS_VALU * 2 / (kernel_cycles * 1024 SIMDs)
   valu_lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
   valu_useful_frac     = valu_issue_frac * valu_lane_utilisation
   wave time            = SQ_ACTIVE_INST_ANY (issuing) + SQ_WAIT_INST_ANY (issue-stalled) + SQ_WAIT_ANY (waiting) ~ SQ_WAVE_CYCLES"""
import collections, csv, glob, json, sys

d, out, scene, W, H, spp = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
tot = collections.OrderedDict()
kernel, ms, n = None, [], 0
for f in sorted(glob.glob(d + "/p*/r_counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "qa_integrate" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            key = (r["Dispatch_Id"] if "Dispatch_Id" in r else r["Start_Timestamp"])
            if key not in seen:
                seen.add(key)
                ms.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
            vg, sc = r["VGPR_Count"], r.get("Scratch_Size", "")
g = tot.get
cyc = g("SQ_BUSY_CYCLES", 0) / 32.0
der = {}
if cyc and g("SQ_INSTS_VALU"):
    der["kernel_cycles"] = cyc
    der["valu_issue_frac"] = g("SQ_INSTS_VALU") * 2.0 / (cyc * 1024)
if g("SQ_ACTIVE_INST_VALU"):
    der["valu_lane_utilisation"] = g("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * g("SQ_ACTIVE_INST_VALU"))
if "valu_issue_frac" in der and "valu_lane_utilisation" in der:
    der["valu_useful_frac"] = der["valu_issue_frac"] * der["valu_lane_utilisation"]
if g("SQ_WAVE_CYCLES"):
    der["wave_time_issuing"] = g("SQ_ACTIVE_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    der["wave_time_issue_stalled"] = g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")
    der["wave_time_waiting"] = g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")
    der["wave_time_lds_issue_stall"] = g("SQ_WAIT_INST_LDS", 0) / g("SQ_WAVE_CYCLES")
if g("SQ_INSTS_VMEM_WR") is not None:
    der["vector_store_bytes_per_launch(64 lanes x 4 B)"] = g("SQ_INSTS_VMEM_WR", 0) * 256.0
json.dump({"scene": scene, "frame": [W, H], "spp": spp, "kernel": kernel, "kernel_ms_per_pass": ms, "vgpr_field": vg, "scratch_bytes_per_lane": sc,
           "counters": tot, "derived": der, "how": __doc__}, open(out, "w"), indent=1)
print(json.dumps(der, indent=1))
