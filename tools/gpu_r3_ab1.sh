#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3_ab1
mkdir -p $OUT
bash $R/tools/gpu_libs_cs.sh --spp 16 lib lib_t2 lib_t3 > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
