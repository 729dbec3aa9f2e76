#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats + HBM PMC passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT/prof
nproc > $OUT/host_info.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/host_info.txt 2>/dev/null; python3 -c "import os;print(len(os.sched_getaffinity(0)))" >> $OUT/host_info.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $OUT/pytest_gpu.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $OUT/smoke.log || exit 1
timeout -k 10 600 python bench.py 2>&1 | tee $OUT/bench.log || exit 1
export TMPDIR=/tmp
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/stats -o r01 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-spp 0 > $OUT/prof/stats.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof/pmc_fetch -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-spp 0 > $OUT/prof/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof/pmc_write -o r01 -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-spp 0 > $OUT/prof/pmc_write.log 2>&1 || exit 1
ls -R $OUT/prof | head -50
