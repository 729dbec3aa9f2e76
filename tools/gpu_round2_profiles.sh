#!/bin/bash
# Round-2 evidence run: bench lines at BASELINE's own sizes, rocprofv3 kernel stats, HBM traffic and SQ passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/round02
mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 scenes/gen_assets.py > /dev/null
# 1. bench lines (default workload first, then the other configs at their BASELINE spp)
python3 bench.py --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --config c3 --steps 2 --warmup 1 --cpu-spp 2 > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --config c4 --steps 1 --warmup 1 --cpu-spp 1 > $O/bench_c4.json 2> $O/bench_c4.err
echo "c2-c4 done" > $O/progress.txt
python3 bench.py --config c5 --steps 1 --warmup 1 --cpu-spp 1 > $O/bench_c5.json 2> $O/bench_c5.err
echo "c5 done" >> $O/progress.txt
python3 bench.py --config c5 --steps 1 --warmup 0 --cpu-spp 0 --spp 256 --pipeline mega > $O/bench_c5_mega_256spp.json 2> /dev/null
python3 bench.py --config c5 --steps 1 --warmup 0 --cpu-spp 0 --spp 256 --pipeline staged > $O/bench_c5_staged_256spp.json 2> /dev/null
python3 bench.py --config c3 --steps 1 --warmup 0 --cpu-spp 0 --pipeline staged > $O/bench_c3_staged.json 2> /dev/null
QA_WF_GROUPS=1 python3 bench.py --config c5 --steps 1 --warmup 0 --cpu-spp 0 --spp 256 --pipeline staged > $O/bench_c5_staged_256spp_1group.json 2> /dev/null
python3 bench.py --config c4 --steps 1 --warmup 0 --cpu-spp 0 --spp 128 --pipeline staged > $O/bench_c4_staged_128spp.json 2> /dev/null
echo "bench done" >> $O/progress.txt
# 2. kernel stats + HBM traffic: C2 default, C5 staged at 64 spp
OUT=gpurun_out/round02/prof_c2 ARGS="--steps 2 --warmup 1 --cpu-spp 0" tools/gpu_pmc_bench.sh > $O/prof_c2.txt 2>&1
OUT=gpurun_out/round02/prof_c5 ARGS="--config c5 --spp 64 --steps 1 --warmup 1 --cpu-spp 0 --pipeline staged" tools/gpu_pmc_bench.sh > $O/prof_c5.txt 2>&1
echo "prof done" >> $O/progress.txt
export PMC_TIMEOUT=200
# 3. SQ / cache counters of the stage kernels (C5 and C3, staged) and of the megakernel on C3
QA_PIPELINE=staged OUT=gpurun_out/round02/pmc_c5_staged tools/gpu_pmc_cmd.sh tools/gpu_one.py trc_scene_tower.xml 3840 2160 32 > $O/pmc_c5_staged.txt 2>&1
QA_PIPELINE=staged OUT=gpurun_out/round02/pmc_c3_staged tools/gpu_pmc_cmd.sh tools/gpu_one.py example_project7_object.xml 1920 1080 32 > $O/pmc_c3_staged.txt 2>&1
QA_PIPELINE=mega OUT=gpurun_out/round02/pmc_c3_mega tools/gpu_pmc_cmd.sh tools/gpu_one.py example_project7_object.xml 1920 1080 32 > $O/pmc_c3_mega.txt 2>&1
QA_PIPELINE=mega OUT=gpurun_out/round02/pmc_c5_mega tools/gpu_pmc_cmd.sh tools/gpu_one.py trc_scene_tower.xml 3840 2160 32 > $O/pmc_c5_mega.txt 2>&1
echo "pmc done" >> $O/progress.txt
timeout -k 10 400 python3 -m pytest tests -m gpu -q -rs > $O/gpu_tests.txt 2>&1
echo "all done" >> $O/progress.txt
