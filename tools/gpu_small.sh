#!/bin/bash
# small-scan solver changes: registrar tests, the C4 loop, the bench's Q-pipe / single-scan legs.  usage: bash tools/gpu_small.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_gpu_voxel.py tests/test_golden.py tests/test_gpu_full.py -m gpu -x -q -k "not legacy and not packed48" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
timeout 900 python bench_c4.py --frames 300 --cpu-frames 100 > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_c4.json gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms","ms_per_frame","ms_per_frame_by_stage","parity_vs_cpu")})
        for k in ("q_pipe",):
            if d.get(k): print("   ", k, {kk:d[k][kk] for kk in d[k] if kk in ("value","ms_per_step","scans_per_s_this_rank")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
