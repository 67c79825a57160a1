#!/bin/bash
# persistence v2 + tile occupancy variants + Q-pipe trace.  usage: bash tools/gpu_r4h.sh <tag>
TAG=${1:-r4h}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_ref_c2.py -m gpu -x -q -k "persistence or tile or matches_oracle or c2 or reuse or determinism or duplicate or subsampl" 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
run() { # name, env..., 
  local name=$1; shift
  env "$@" timeout 300 python bench.py $C > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err
  python - gpurun_out/${TAG}_bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step")})
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
run persist X=1
run nopersist LL_DEBUG_OR=4096
run w6 LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_w6.so
run w8 LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_w8.so
bash tools/gpu_timing.sh $TAG
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --q-pipe > /tmp/prof_$TAG/bench.json 2> /tmp/prof_$TAG/trace.log
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_qpipe_kernel_trace_by_grid.csv
tail -1 /tmp/prof_$TAG/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('qpipe', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
head -30 gpurun_out/${TAG}_qpipe_kernel_trace_by_grid.csv | cut -c1-160
