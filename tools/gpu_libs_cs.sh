#!/bin/bash
# A/B of library BUILDS on the non-resident scenes (one process per build: the library is loaded once per process):
#   tools/gpu_libs_cs.sh [--spp N] [--cases c3,c5] lib lib_t2 ...     (directories under qaray_amd/)
# equal "small-frame sha1" = bit-identical rgb / depth / sample counts of the 1/8-size frame
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS=""
while [[ "$1" == --* ]]; do ARGS="$ARGS $1 $2"; shift 2; done
for d in "$@"; do
  QA_HIP_LIB=$R/qaray_amd/$d/libqaray_hip.so timeout -k 10 600 python3 $R/tools/gpu_ab.py $ARGS $d: 2>&1 | grep -v "^$" || echo "$d FAILED"
done
