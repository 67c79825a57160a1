#!/bin/bash
# the other BASELINE configurations on the current build: Q-pipe, C3, C4 (long sequence), C5.  usage: bash tools/gpu_configs.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench_c4.py --frames 2000 --distinct-frames 100 --cpu-frames 0 > gpurun_out/${TAG}_bench_c4_2000frames.json 2> gpurun_out/${TAG}_c4.err
timeout 600 python bench_c4.py --frames 200 --distinct-frames 100 --cpu-frames 8 > gpurun_out/${TAG}_bench_c4_200frames.json 2>> gpurun_out/${TAG}_c4.err
timeout 600 python bench_c3.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_c3.err
timeout 600 python bench_c5.py > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_c5.err
timeout 600 python bench_c5.py --f16 > gpurun_out/${TAG}_bench_c5_f16.json 2>> gpurun_out/${TAG}_c5.err
for f in c4_2000frames c4_200frames c3 c5 c5_f16; do echo $f; head -c 1500 gpurun_out/${TAG}_bench_$f.json; echo; done
tail -3 gpurun_out/${TAG}_c4.err gpurun_out/${TAG}_c3.err gpurun_out/${TAG}_c5.err
