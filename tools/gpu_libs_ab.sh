#!/bin/bash
# bench.py (default C2 workload) on several builds of libqaray_hip.so: tools/gpu_libs_ab.sh lib lib_w4r lib_w5l ...
# each with its WRITE_SIZE per launch (scratch evictions show up there)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for d in "$@"; do
  export QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so
  python3 $R/bench.py --steps 3 --warmup 1 --cpu-spp 0 > /tmp/b_$d.log 2>&1
  v=$(grep -h '^{' /tmp/b_$d.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f Msamples/s, kernel %.2f ms' % (d['value'], d['roofline']['kernel_ms_avg']))")
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$d -o r -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-spp 0 > /tmp/w_$d.log 2>&1)
  w=$(python3 - <<PY
import csv
tot=0;n=0;sc=''
for r in csv.DictReader(open('/tmp/pw_$d/r_counter_collection.csv')):
    if 'qa_integrate' in r['Kernel_Name']:
        tot+=float(r['Counter_Value']);n+=1;sc=r.get('Scratch_Size','')
print('WRITE_SIZE %.1f MiB/launch, scratch %s B/lane' % (tot/1024/max(n,1), sc))
PY
)
  echo "$d: $v, $w"
done
