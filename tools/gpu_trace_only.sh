#!/bin/bash
# bench + kernel trace only.  usage: bash tools/gpu_trace_only.sh <tag> [extra bench args]
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed "$@" > gpurun_out_$TAG.json 2> /tmp/prof_$TAG/trace.log
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" > gpurun_out/${TAG}_kernel_trace_by_grid.csv
head -24 gpurun_out/${TAG}_kernel_trace_by_grid.csv
