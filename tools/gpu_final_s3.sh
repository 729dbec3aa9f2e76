#!/bin/bash
# End-of-session evidence: bench lines of the BASELINE configs at their own settings and the kernel stats / HBM traffic
# passes of the default bench line (-> tools/make_round_summary.py -> profiles/round02/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_s3
mkdir -p $O
export TMPDIR=/tmp
cd $R
python3 scenes/gen_assets.py > /dev/null
python3 bench.py --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err
python3 bench.py --config c3 --steps 2 --warmup 1 --cpu-spp 0 > $O/bench_c3.json 2> /dev/null
python3 bench.py --config c4 --steps 1 --warmup 1 --cpu-spp 0 > $O/bench_c4.json 2> /dev/null
echo "c2-c4 done" > $O/progress.txt
python3 bench.py --config c5 --steps 1 --warmup 1 --cpu-spp 0 > $O/bench_c5.json 2> /dev/null
echo "c5 done" >> $O/progress.txt
python3 bench.py --config c5 --steps 1 --warmup 0 --cpu-spp 0 --spp 256 --pipeline mega > $O/bench_c5_mega_256spp.json 2> /dev/null
python3 bench.py --config c3 --steps 1 --warmup 0 --cpu-spp 0 --pipeline staged > $O/bench_c3_staged.json 2> /dev/null
python3 bench.py --config c4 --steps 1 --warmup 0 --cpu-spp 0 --spp 128 --pipeline staged > $O/bench_c4_staged_128spp.json 2> /dev/null
echo "bench done" >> $O/progress.txt
OUT=gpurun_out/final_s3/prof_c2 ARGS="--steps 2 --warmup 1 --cpu-spp 0" tools/gpu_pmc_bench.sh > $O/prof_c2.txt 2>&1
echo "prof done" >> $O/progress.txt
for f in $O/bench_*.json; do python3 -c "import json,sys; d=json.load(open('$f')); print('$(basename $f)', round(d['value'],1), d['unit'], d['roofline']['kernel'][:60], round(d['ms_per_step'],1), 'ms')"; done
