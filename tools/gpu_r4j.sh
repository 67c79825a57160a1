#!/bin/bash
# batched hash inserts / de-duplication marks: registration tests, key-frame exploration, bench, timers.  usage: bash tools/gpu_r4j.sh <tag>
TAG=${1:-r4j}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_reg.py tests/test_ref_c2.py tests/test_golden.py tests/test_ref_golden.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
timeout 400 python tools/exp_keyframes.py > gpurun_out/${TAG}_keyframes.txt 2>&1; tail -20 gpurun_out/${TAG}_keyframes.txt | cut -c1-700
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
bash tools/gpu_timing.sh $TAG
