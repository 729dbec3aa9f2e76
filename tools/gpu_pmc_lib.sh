#!/bin/bash
# SQ counter passes of one scene on one library build: LIB=lib_old SCENE=... SPP=.. W=.. H=.. tools/gpu_pmc_lib.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
LIB=${LIB:-lib}
OUT=$R/gpurun_out/pmc_$LIB
mkdir -p $OUT
python3 $R/scenes/gen_assets.py > /dev/null
export TMPDIR=/tmp
export QA_HIP_LIB=$R/qaray_amd/$LIB/libqaray_hip.so
cd /tmp
ARGS="--no-other-configs --steps 1 --warmup 0 --cpu-spp 0 --spp ${SPP:-8} --scene ${SCENE:-trc_scene_tower.xml} --width ${W:-1920} --height ${H:-1080}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_FLAT SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32" ; do
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
            tot["_kernel_ms(" + f.split("/")[-2] + ")"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            tot["_vgpr"] = r["VGPR_Count"]; tot["_scratch"] = r.get("Scratch_Size", "")
print("$LIB:", " ".join(f"{k}={v:.4g}" if isinstance(v, float) else f"{k}={v}" for k, v in tot.items()))
PY
grep -h "^{" $OUT/p1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'Msamples/s', d['config']['casts_per_sample'], 'casts/sample')"
