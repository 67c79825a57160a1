#!/bin/bash
# the round's committed bench lines: the default bench.py line and the C4 mapping loop with every frame distinct.  usage: bash tools/gpu_final_bench.sh <tag> [frames] [cpu frames]
TAG=${1:-x}; FR=${2:-2000}; CF=${3:-500}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 1500 python bench_c4.py --frames $FR --cpu-frames $CF > gpurun_out/${TAG}_bench_c4_${FR}frames.json 2> gpurun_out/${TAG}_bench_c4_${FR}frames.err
tail -c 2500 gpurun_out/${TAG}_bench_default.json; echo
tail -c 1500 gpurun_out/${TAG}_bench_c4_${FR}frames.json; tail -3 gpurun_out/${TAG}_bench_c4_${FR}frames.err
