#!/bin/bash
# HBM traffic of bench.py's workload: separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass),
# kernel-trace stats in a third.  ARGS="--config c5 --spp 64" OUT=gpurun_out/prof_x tools/gpu_pmc_bench.sh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${OUT:-gpurun_out/prof}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS=${ARGS:-"--steps 2 --warmup 1 --cpu-spp 0"}
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $R/bench.py $ARGS > $OUT/bench.log 2>&1 || { echo "stats pass failed"; tail -3 $OUT/bench.log; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- python3 $R/bench.py $ARGS > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -3 $OUT/fetch.log; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- python3 $R/bench.py $ARGS > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -3 $OUT/write.log; }
python3 - <<PY
import csv, collections
for name in ("fetch", "write"):
    tot = collections.defaultdict(lambda: [0.0, 0])
    try:
        for r in csv.DictReader(open("$OUT/pmc_%s/r_counter_collection.csv" % name)):
            k = r["Kernel_Name"].split("(")[0][-40:]
            tot[k][0] += float(r["Counter_Value"]); tot[k][1] += 1
            scratch = r.get("Scratch_Size", "")
            tot[k].append(scratch) if len(tot[k]) == 2 else None
    except FileNotFoundError:
        continue
    for k, v in tot.items():
        print(f"{name.upper()}_SIZE {k:42s} total {v[0] / 1024:12.1f} MiB over {v[1]} launches = {v[0] / 1024 / v[1]:10.2f} MiB/launch  scratch/lane {v[2] if len(v) > 2 else '?'}")
PY
grep -h '^{' $OUT/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['unit'], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'])"
