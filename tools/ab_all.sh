#!/bin/bash
# A/B several builds of libqaray_hip on the bench scene AND the big-scene probe in one GPU visit:
#   tools/ab_all.sh lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in "$@"; do
  echo "== $lib"
  QA_HIP_LIB=$R/qaray_amd/lib/$lib timeout -k 10 300 python $R/bench.py --cpu-spp 0 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')" || exit 1
  QA_HIP_LIB=$R/qaray_amd/lib/$lib timeout -k 10 300 python $R/tools/gpu_bigscenes.py 2>&1 | grep "x" || exit 1
done
