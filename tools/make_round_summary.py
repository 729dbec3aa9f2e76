"""Turn one tools/gpu_pmc_bench.sh output directory into the committed evidence of a bench line:
   python tools/make_round_summary.py gpurun_out/round02/prof_c5 profiles/round02/c5_staged_64spp
writes <prefix>_summary.json (what bench.py's `traffic` lookup reads), <prefix>_kernel_stats.csv (rocprofv3 --kernel-trace
--stats), <prefix>_pmc_fetch.csv / _pmc_write.csv (per-kernel sums of the separate --pmc FETCH_SIZE / WRITE_SIZE passes)
and <prefix>_bench.log.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md: KiB units; gfx950
FETCH_SIZE counts 64-byte halves of the 128-byte fetches), per bench "launch" (= one frame for the staged integrator)."""
import collections, csv, json, os, shutil, subprocess, sys
src, prefix = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(prefix), exist_ok=True)
line = [l for l in open(os.path.join(src, "bench.log")) if l.startswith("{")][0]
bench = json.loads(line)
frames = bench["steps"] + bench["warmup"]
product = lambda k: k.startswith("qa::") or "qa_integrate" in k or "qa::" in k
sums = {}
for name in ("fetch", "write"):
    tot = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(src, "pmc_" + name, "r_counter_collection.csv"))):
        k = r["Kernel_Name"]
        t = tot.setdefault(k, [0.0, 0, r.get("VGPR_Count", ""), r.get("Scratch_Size", ""), r.get("LDS_Block_Size", "")])
        t[0] += float(r["Counter_Value"]); t[1] += 1
    with open(prefix + "_pmc_%s.csv" % name, "w") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value_sum_KiB", "Launches", "VGPR_Count", "Scratch_Size", "LDS_Block_Size"])
        for k, t in tot.items():
            w.writerow([k, name.upper() + "_SIZE", t[0], t[1], t[2], t[3], t[4]])
    sums[name] = sum(t[0] for k, t in tot.items() if product(k))
shutil.copy(os.path.join(src, "stats", "r_kernel_stats.csv"), prefix + "_kernel_stats.csv")
open(prefix + "_bench.log", "w").write(line)
kstats = {}
for r in csv.DictReader(open(prefix + "_kernel_stats.csv")):
    if product(r["Name"]):
        kstats[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "total_ms": int(r["TotalDurationNs"]) / 1e6}
try:
    build = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    build = "?"
summary = {
    "scene": os.path.basename(bench["config"]["workload"].split(" ")[0]), "frame": bench["config"]["frame"], "spp": bench["config"]["spp"],
    "kernel_name": bench["roofline"]["kernel"].split(" (")[0], "build": build,
    "bench_value": bench["value"], "bench_unit": bench["unit"], "bench_kernel_ms_avg": bench["roofline"]["kernel_ms_avg"],
    "frames_in_each_pmc_pass": frames,
    "fetch_size_kib_per_launch": sums["fetch"] / frames, "write_size_kib_per_launch": sums["write"] / frames,
    "hbm_traffic_bytes_per_launch": (2.0 * sums["fetch"] + sums["write"]) * 1024.0 / frames,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "rocprof_kernel_stats": kstats,
    "rocprof_ms_per_launch": sum(v["total_ms"] for v in kstats.values()) / frames,
}
json.dump(summary, open(prefix + "_summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
