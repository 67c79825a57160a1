#!/bin/bash
# line-search changes: the bounded / rejected / sub-sampled registrations against the oracle, then the bench's Q-pipe leg.  usage: bash tools/gpu_bounded.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_reg.py tests/test_gpu_voxel.py tests/test_ref_c2.py tests/test_ref_golden.py -m gpu -x -q -k "bounded or reject or subsampl or downsampled or ref_c2 or ref_golden" 2>&1 | tail -8 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -3 gpurun_out/${TAG}_tests.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms")}, d.get("q_pipe"))
PY
