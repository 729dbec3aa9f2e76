#!/bin/bash
# Timing ablations (NOT parity builds): bench.py C2 on several builds of libqaray_hip.so and env settings.
#   tools/gpu_ablate.sh "lib lib_fdiv lib_fdsc" "QA_SYNC=0"      (second argument: env settings tried on the first lib)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
one() {  # name, env...
  local name=$1; shift
  env "$@" python3 $R/bench.py --steps 3 --warmup 1 --cpu-spp 0 > /tmp/b_$name.log 2>&1
  grep -h '^{' /tmp/b_$name.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$name: %.0f Msamples/s, kernel %.2f ms [%s]' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel']))" || tail -n 5 /tmp/b_$name.log
}
for d in $1; do one $d QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so; done
first=$(echo $1 | cut -d' ' -f1)
for e in $2; do one "${first}_$e" QA_HIP_LIB=$R/qaray_amd/$first/libqaray_hip.so $e; done
