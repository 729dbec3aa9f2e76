#!/bin/bash
# SQ / cache counter passes on an arbitrary python command, summed per kernel:
#   OUT=gpurun_out/pmc_x tools/gpu_pmc_cmd.sh tools/gpu_one.py trc_scene_tower.xml 3840 2160 16
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${OUT:-gpurun_out/pmc_cmd}
mkdir -p $OUT
python3 $R/scenes/gen_assets.py > /dev/null
export TMPDIR=/tmp
cd /tmp
SCRIPT=$R/$1; shift
i=0
# SETS="A B C;D E" replaces the default counter sets (one pass per set)
if [ -n "$SETS" ]; then IFS=';' read -ra CUSTOM <<< "$SETS"; else CUSTOM=(); fi
for set in "${CUSTOM[@]}"; do
  i=$((i+1))
  timeout -k 10 ${PMC_TIMEOUT:-400} rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
[ -n "$SETS" ] && PMC_SETS=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ; do
  i=$((i+1))
  [ -n "$PMC_SETS" ] && { [ "$PMC_SETS" = 0 ] || [ $i -gt $PMC_SETS ]; } && break
  timeout -k 10 ${PMC_TIMEOUT:-400} rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.OrderedDict())
for f in sorted(glob.glob("$OUT/p*/r_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-24:]
        d = tot[k]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
        d["_vgpr"] = r["VGPR_Count"]; d["_lds"] = r["LDS_Block_Size"]; d["_scratch"] = r.get("Scratch_Size", "")
for k, d in tot.items():
    print("==", k)
    for n, v in d.items():
        print(f"   {n:34s} {v}")
    g = d.get
    if g("SQ_WAVE_CYCLES"):
        print(f"   -> wait_any/wave_cycles {g('SQ_WAIT_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}  issue-stall {g('SQ_WAIT_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}  active {g('SQ_ACTIVE_INST_ANY', 0) / g('SQ_WAVE_CYCLES'):.3f}")
    if g("SQ_ACTIVE_INST_VALU"):
        print(f"   -> lane utilisation (thread_cycles / 64 / active_inst_valu) {g('SQ_THREAD_CYCLES_VALU', 0) / 64 / g('SQ_ACTIVE_INST_VALU'):.3f}")
    if g("SQ_BUSY_CYCLES"):
        print(f"   -> VALU wave-instr per SIMD busy cycle {g('SQ_INSTS_VALU', 0) / g('SQ_BUSY_CYCLES'):.4f} (busy cycles are per SE? see guide)")
    if g("TCC_REQ_sum"):
        print(f"   -> L2 hit rate {g('TCC_HIT_sum', 0) / max(1.0, g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0)):.3f}")
PY
grep -h "Msamples" $OUT/p1.log
