#!/bin/bash
# round 6: kernel trace of the one-batch-at-a-time default loop only (quick look between changes).  usage: bash tools/gpu_r6_trace.sh <tag> [bench args]
TAG=${1:-r06a}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
P=/tmp/prof_$TAG; rm -rf $P; mkdir -p $P; cd /tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed --no-pipeline "$@" > $P/trace.log 2>&1
cd "$R"
python tools/summarize_rocprof.py trace "$(find $P/trace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
head -12 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -c1-200
