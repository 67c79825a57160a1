#!/bin/bash
# targeted tests + bench + kernel trace.  usage: bash tools/gpu_trace.sh <tag> [pytest files...]
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
FILES=${@:-tests/test_gpu_reg.py tests/test_gpu_full.py tests/test_mapping_sequence.py tests/test_golden.py tests/test_ref_golden.py}
( timeout 1200 python -m pytest $FILES -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > /tmp/prof_$TAG/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
tail -4 gpurun_out/${TAG}_tests.log
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print({k:d[k] for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms","knn_reuse_last_iter")})
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-3000:])
PY
head -16 gpurun_out/${TAG}_kernel_trace_by_grid.csv
