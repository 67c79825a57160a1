#!/bin/bash
# kernel timeline of the pipelined loop: how busy is the device, how much do the two batches overlap.  usage: bash tools/gpu_r4r.sh <tag>
TAG=${1:-r4r}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for d in 2 4; do
rm -rf /tmp/tr_$TAG$d
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$TAG$d -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed --in-flight $d > /tmp/tr_$TAG$d.log 2>&1
T=$(find /tmp/tr_$TAG$d -name '*kernel_trace.csv' | head -1)
tail -1 /tmp/tr_$TAG$d.log | cut -c1-200
python $GRAFT_REPO_ROOT/tools/trace_overlap.py "$T" 50 | tee $GRAFT_REPO_ROOT/gpurun_out/${TAG}_overlap_d$d.txt
done
