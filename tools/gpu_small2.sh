#!/bin/bash
# small-scan solver changes: registrar / voxel / mapping tests, solver phase timers on Q-pipe scans, C4 loop.  usage: bash tools/gpu_small2.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_gpu_voxel.py tests/test_golden.py tests/test_gpu_full.py -m gpu -x -q -k "not legacy and not packed48" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log
bash tools/gpu_qtiming.sh $TAG
timeout 900 python bench_c4.py --frames 600 --cpu-frames 100 > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
python - gpurun_out/${TAG}_c4.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","ms_per_frame","ms_per_frame_by_stage","parity_vs_cpu")})
PY
