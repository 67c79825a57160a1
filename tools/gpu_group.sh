#!/bin/bash
# grouped-solver bring-up: a few B=1 tests under a short timeout first, then the registration tests, then latency A/B
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 300 python -m pytest tests/test_golden.py tests/test_ref_golden.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_t1.log 2>&1
tail -6 gpurun_out/${TAG}_t1.log
if grep -q "passed" gpurun_out/${TAG}_t1.log && ! grep -q "failed\|error" gpurun_out/${TAG}_t1.log; then
  ( timeout 1200 python -m pytest tests/test_gpu_reg.py tests/test_gpu_full.py tests/test_mapping_sequence.py tests/test_adapter_cpp.py tests/test_ll_node.py tests/test_gpu_multigpu.py -m gpu -q 2>&1 | tail -40 ) > gpurun_out/${TAG}_t2.log 2>&1
  tail -30 gpurun_out/${TAG}_t2.log
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > gpurun_out/${TAG}_bench_group.json 2> gpurun_out/${TAG}_bench.err
  timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed --no-solver-groups > gpurun_out/${TAG}_bench_nogroup.json 2>> gpurun_out/${TAG}_bench.err
  for f in group nogroup; do python - gpurun_out/${TAG}_bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print({k:d[k] for k in ("value","single_scan_latency_ms","single_scan_solver")})
except Exception as e: print("ERR",e)
PY
  done
fi
