#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQ_INST_LEVEL[A-Z_0-9]*\|SQ_WAIT_IFETCH[A-Z_0-9]*\|SQ_LEVEL_WAVES\|SQ_ACCUM_PREV[A-Z_0-9]*\|SQ_INSTS_BRANCH\|SQ_INSTS_CBRANCH[A-Z_]*\|SQ_VALU_[A-Z_0-9]*\|SQ_INSTS_VALU_[A-Z_0-9]*\|SQ_WAVES_EQ_64\|SQ_WAVES_LT[_0-9]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
cat $OUT/avail.txt
ARGS="--steps 1 --warmup 0 --cpu-spp 0 --spp 128"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INSTS_CBRANCH_TAKEN SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $R/bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/r_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "qa_integrate" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, v in tot.items():
    print(f"{k:34s} {v:.6g}")
PY
