#!/bin/bash
# round 4's committed bench lines on the final build: bench.py default (all legs), C4 (2000 distinct frames, 500 through the oracle), C3, C5.
TAG=${1:-r04a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
timeout 900 python bench_c4.py --frames 2000 --cpu-frames 500 > gpurun_out/${TAG}_bench_c4_2000frames.json 2> gpurun_out/${TAG}_bench_c4_2000frames.err
timeout 600 python bench_c3.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
timeout 600 python bench_c5.py > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
timeout 600 python bench_c5.py --f16 > gpurun_out/${TAG}_bench_c5_f16.json 2> gpurun_out/${TAG}_bench_c5_f16.err
for f in bench_default bench_c4_2000frames bench_c3 bench_c5 bench_c5_f16; do echo "== $f"; tail -c 2800 gpurun_out/${TAG}_$f.json; echo; tail -2 gpurun_out/${TAG}_$f.err | cut -c1-300; done
