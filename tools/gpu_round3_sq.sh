#!/bin/bash
# SQ counter passes of BASELINE C3 / C4 / C5 at their own spp (three rocprofv3 --pmc passes each, tools/gpu_pmc_lib.sh) -> gpurun_out/pmc_<cfg>
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
run() { LIB=lib SCENE=$2 W=$3 H=$4 SPP=$5 bash tools/gpu_pmc_lib.sh > gpurun_out/sq_$1.log 2>&1; rm -rf gpurun_out/pmc_$1; mv gpurun_out/pmc_lib gpurun_out/pmc_$1; tail -2 gpurun_out/sq_$1.log; }
run c3 example_project7_object.xml 1920 1080 256
run c4 example_project12_caustics_glossy.xml 3840 2160 1024
run c5 trc_scene_tower.xml 3840 2160 2048
