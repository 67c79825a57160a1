#!/bin/bash
# one-workgroup voxel filter: its tests, the mapping-loop tests that run on it, then the C4 loop.  usage: bash tools/gpu_vox.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_full.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
timeout 900 python bench_c4.py --frames 300 --cpu-frames 100 > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
LL_VOXEL_GENERAL_PATH=1 timeout 900 python bench_c4.py --frames 300 --cpu-frames 0 > gpurun_out/${TAG}_c4_general.json 2> gpurun_out/${TAG}_c4_general.err
python - gpurun_out/${TAG}_c4.json gpurun_out/${TAG}_c4_general.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_frame","ms_per_frame_by_stage","frames_per_sequence","parity_vs_cpu")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
