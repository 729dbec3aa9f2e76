#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one scene on several library builds:
#   SCENE=example_project12_caustics_glossy.xml W=3840 H=2160 SPP=16 tools/gpu_hbm_ab.sh lib lib_r2
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
python3 $R/scenes/gen_assets.py > /dev/null
cd /tmp
for d in "$@"; do
  export QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so
  for ctr in FETCH_SIZE WRITE_SIZE; do
    OUT=$R/gpurun_out/hbm_${d}_$ctr
    mkdir -p $OUT
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $OUT -o r -- python3 $R/tools/gpu_one.py ${SCENE:-trc_scene_tower.xml} ${W:-3840} ${H:-2160} ${SPP:-16} > $OUT/log.txt 2>&1 || { echo "$d $ctr failed"; tail -3 $OUT/log.txt; }
    python3 - <<PY
import csv
tot, ms, name = 0.0, 0.0, ""
for r in csv.DictReader(open("$OUT/r_counter_collection.csv")):
    if "qa_integrate" in r["Kernel_Name"] and r["Counter_Name"] == "$ctr":
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
        if t > ms: ms, tot, name = t, float(r["Counter_Value"]), r["Kernel_Name"][:40]   # the full-frame launch
mult = 2 if "$ctr" == "FETCH_SIZE" else 1   # gfx950: FETCH_SIZE counts 64-byte halves of 128-byte requests once (guide); both in KiB
print(f"$d $ctr: {tot * 1024 * mult / 1e9:.3f} GB in the {ms:.1f} ms launch of {name}")
PY
  done
done
