#!/bin/bash
# Kernel timeline of one staged frame: rocprofv3 --kernel-trace of tools/gpu_one.py, per-dispatch start / end timestamps.
#   tools/gpu_timeline.sh <tag> <scene.xml> <w> <h> <spp>        (QA_WF_GROUPS etc. from the environment)  -> gpurun_out/timeline_<tag>.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
tag=$1; shift
cd /tmp && QA_PIPELINE=staged rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o t -- python3 $R/tools/gpu_one.py "$@" > /tmp/tl_$tag.log 2>&1
tail -n 2 /tmp/tl_$tag.log
f=$(find /tmp/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$R/gpurun_out/timeline_$tag.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
out.write("kernel,queue,start_ns,end_ns\n")
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]
    short = "logic" if "wf_logic" in n else "cull" if "wf_cull" in n else "trace" if "wf_trace" in n else "redo" if "wf_redo" in n else "init" if "wf_init" in n else n[:24]
    out.write(f"{short},{r.get('Queue_Id','')},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0}\n")
print(len(rows), "dispatches")
PY
