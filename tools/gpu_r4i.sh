#!/bin/bash
# butterfly reduction + persistence v3: full registration tests, key-frame tests, bench A/B, timers.  usage: bash tools/gpu_r4i.sh <tag>
TAG=${1:-r4i}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_reg.py tests/test_ref_c2.py tests/test_golden.py tests/test_ref_golden.py tests/test_keyframes.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py $C > gpurun_out/${TAG}_bench_$name.json 2> gpurun_out/${TAG}_bench_$name.err
  python - gpurun_out/${TAG}_bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms")})
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
run persist X=1
run nopersist LL_DEBUG_OR=4096
bash tools/gpu_timing.sh $TAG
LL_DEBUG_OR=4096 bash tools/gpu_timing.sh ${TAG}_nopersist
