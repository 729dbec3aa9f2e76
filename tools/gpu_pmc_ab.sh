#!/bin/bash
# Core SQ counters of the bench kernel for two kernel families: tools/gpu_pmc_ab.sh lockstep wg
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --cpu-spp 0 --spp ${SPP:-128}"
for k in "$@"; do
  OUT=$R/gpurun_out/pmc_ab_$k
  mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_FLAT"; do
    i=$((i+1))
    QA_KERNEL=$k timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  done
  echo "== $k"
  python3 - <<PY
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/r_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "qa_integrate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            tot["_ms(" + f.split("/")[-2] + ")"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            tot["_scratch"] = r.get("Scratch_Size", ""); tot["_lds"] = r["LDS_Block_Size"]; tot["_grid"] = r["Grid_Size"]
for k, v in tot.items():
    print(f"{k:28s} {v}")
t = tot
print("lane utilisation %.3f  valu-active/wave-cycles %.3f  wait/wave-cycles %.3f" % (t["SQ_THREAD_CYCLES_VALU"] / (64 * t["SQ_ACTIVE_INST_VALU"]), t["SQ_ACTIVE_INST_VALU"] / t["SQ_WAVE_CYCLES"], t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]))
PY
done
