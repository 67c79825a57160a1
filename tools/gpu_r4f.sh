#!/bin/bash
# tests (k-NN, key frames, cell maps) + default bench + trace + solver phase timers.  usage: bash tools/gpu_r4f.sh <tag>
TAG=${1:-r4f}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_keyframes.py tests/test_cellmap.py tests/test_ref_c2.py -m gpu -x -q -k "tile or knn or wavefront or matches_oracle or batch_pipeline or keyframe or touched or loop or cell or c2" 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -12 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 300 python bench.py $C > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms")})
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > /tmp/prof_$TAG/bench.json 2> /tmp/prof_$TAG/trace.log
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
head -6 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -c1-200
bash tools/gpu_timing.sh $TAG
