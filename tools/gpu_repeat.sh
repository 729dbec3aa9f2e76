#!/bin/bash
# Repeat bench.py in fresh processes under several environments: tools/gpu_repeat.sh "<bench args>" N "ENV=.. ENV=.." "ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=$1; N=$2; shift 2
for envs in "$@"; do
  vals=""
  for i in $(seq 1 $N); do
    v=$(env $envs python3 $R/bench.py $ARGS --cpu-spp 0 2>/dev/null | python3 -c "import sys,json; print('%.0f' % json.loads(sys.stdin.readline())['value'])")
    vals="$vals $v"
  done
  echo "[$envs] $vals"
done
