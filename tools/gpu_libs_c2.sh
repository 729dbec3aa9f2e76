#!/bin/bash
# bench.py (C2 only) on several library builds, twice each: tools/gpu_libs_c2.sh lib lib_x ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
for d in "$@"; do
  for rep in 1 2; do
    QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so timeout -k 10 300 python3 $R/bench.py --no-other-configs --cpu-spp 0 --steps 3 --warmup 1 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms')"
  done
done
