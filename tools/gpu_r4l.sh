#!/bin/bash
# the two repaired GPU tests + in-kernel phase timers on small scans (Q-pipe batch of 16, the C4 loop).  usage: bash tools/gpu_r4l.sh <tag>
TAG=${1:-r4l}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_keyframes.py tests/test_cellmap.py -m gpu -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
bash tools/gpu_qtiming.sh $TAG
LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 300 python bench_c4.py --frames 200 --cpu-frames 0 > gpurun_out/${TAG}_c4timing.json 2> gpurun_out/${TAG}_c4timing.err
tail -1 gpurun_out/${TAG}_c4timing.json | cut -c1-1500
