#!/bin/bash
# Round-3 evidence on one MI355X: GPU tests, the default bench line (C2 + other_configs), and per BASELINE config a rocprofv3 kernel-stats
# pass plus separate FETCH_SIZE / WRITE_SIZE passes at the config's own spp.  Outputs under gpurun_out/round03/ (summaries are made from
# them with tools/make_round_summary.py and committed under profiles/round03/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/round03
mkdir -p $R/$O
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout -k 10 600 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default.json; cut -c1-400 $O/bench_default.json
for c in c2 c3 c4 c5; do
  ARGS="--config $c --steps 1 --warmup 0 --cpu-spp 0 --no-other-configs" OUT=$O/prof_$c bash tools/gpu_pmc_bench.sh 2>&1 | tail -4
done
