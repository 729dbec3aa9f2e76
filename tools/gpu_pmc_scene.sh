#!/bin/bash
# SQ/TCP counter passes on an arbitrary scene: SCENE=trc_scene_tower.xml SPP=8 tools/gpu_pmc_scene.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_scene
mkdir -p $OUT
python3 $R/scenes/gen_assets.py > /dev/null
export TMPDIR=/tmp
cd /tmp
ARGS="--no-other-configs --steps 1 --warmup 0 --cpu-spp 0 --spp ${SPP:-8} --scene ${SCENE:-trc_scene_tower.xml}"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" ; do
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
            tot["_vgpr"] = r["VGPR_Count"]; tot["_lds"] = r["LDS_Block_Size"]; tot["_grid"] = r["Grid_Size"]
for k, v in tot.items():
    print(f"{k:34s} {v}")
PY
grep -h "^{" $OUT/p1.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], 'Msamples/s', d['config']['casts_per_sample'], 'casts/sample')"
