#!/bin/bash
# Two SQ counter passes of one scene on several library builds (instruction counts, wave-time split, vector-memory instructions):
#   SCENE=example_project12_caustics_glossy.xml W=3840 H=2160 SPP=16 tools/gpu_sq_ab.sh lib lib_r2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
python3 $R/scenes/gen_assets.py > /dev/null
cd /tmp
for d in "$@"; do
  export QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so
  OUT=$R/gpurun_out/sq_$d
  mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -o r -- python3 $R/tools/gpu_one.py ${SCENE:-trc_scene_tower.xml} ${W:-3840} ${H:-2160} ${SPP:-16} > $OUT/p$i.log 2>&1 || { echo "$d pass $i failed"; tail -3 $OUT/p$i.log; }
  done
  python3 - <<PY
import csv, glob
best = {}
for f in sorted(glob.glob("$OUT/p*/r_counter_collection.csv")):
    rows = [r for r in csv.DictReader(open(f)) if "qa_integrate" in r["Kernel_Name"]]
    if not rows: continue
    tmax = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) == tmax:
            best[r["Counter_Name"]] = best.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
            best["_ms"] = tmax * 1e-6
g = best.get
cyc = g("SQ_BUSY_CYCLES", 0) / 32.0
print("$d: %.1f ms  VALU %.3g  SALU %.3g  LDS %.3g  SMEM %.3g  VMEM_RD %.3g  VMEM_WR %.3g  FLAT %.3g | valu_issue_frac %.3f  lane_util %.3f | wave time: issuing %.3f stalled %.3f waiting %.3f" % (
    g("_ms", 0), g("SQ_INSTS_VALU", 0), g("SQ_INSTS_SALU", 0), g("SQ_INSTS_LDS", 0), g("SQ_INSTS_SMEM", 0), g("SQ_INSTS_VMEM_RD", 0), g("SQ_INSTS_VMEM_WR", 0), g("SQ_INSTS_FLAT", 0),
    g("SQ_INSTS_VALU", 0) * 2.0 / max(1.0, cyc * 1024), g("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, 64.0 * g("SQ_ACTIVE_INST_VALU", 1)),
    g("SQ_ACTIVE_INST_ANY", 0) / max(1.0, g("SQ_WAVE_CYCLES", 1)), g("SQ_WAIT_INST_ANY", 0) / max(1.0, g("SQ_WAVE_CYCLES", 1)), g("SQ_WAIT_ANY", 0) / max(1.0, g("SQ_WAVE_CYCLES", 1))))
PY
done
