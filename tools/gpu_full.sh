#!/bin/bash
# full GPU tier: every -m gpu test, the default bench line, rocprofv3 kernel trace and HBM byte counters (separate passes)
# usage: bash tools/gpu_full.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- $B > /tmp/prof_$TAG/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_$TAG/fetch -- $B > /tmp/prof_$TAG/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_$TAG/write -- $B > /tmp/prof_$TAG/write.log 2>&1
cd "$GRAFT_REPO_ROOT"
T=$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)
F=$(find /tmp/prof_$TAG/fetch -name '*counter_collection.csv' | head -1)
W=$(find /tmp/prof_$TAG/write -name '*counter_collection.csv' | head -1)
python tools/summarize_rocprof.py trace "$T" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
python tools/summarize_rocprof.py codeobj loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_code_objects.csv
python tools/summarize_rocprof.py pmc "$F" "$W" > gpurun_out/${TAG}_pmc_hbm_bytes.csv
cp $(find /tmp/prof_$TAG/trace -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_raw.csv
tail -6 gpurun_out/${TAG}_tests.log
tail -c 3000 gpurun_out/${TAG}_bench.json
head -14 gpurun_out/${TAG}_kernel_trace_by_grid.csv; head -6 gpurun_out/${TAG}_pmc_hbm_bytes.csv
