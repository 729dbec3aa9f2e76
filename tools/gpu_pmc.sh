#!/bin/bash
# SQ counter passes over a short bench run (separate --pmc passes, no tracing domains).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --cpu-spp 0 --spp ${SPP:-128}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/r_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "qa_integrate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            tot["_kernel_ms(" + f.split("/")[-2] + ")"] = dur
for k, v in tot.items():
    print(f"{k:28s} {v:.6g}")
PY
