#!/bin/bash
# Q-pipe with batches in flight; C4 with the next frame extracted during registration (A/B); mapping sequence tests.  usage: bash tools/gpu_r4p.sh <tag>
TAG=${1:-r4p}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench_c4.py --frames 400 --cpu-frames 40 > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
timeout 300 python bench_c4.py --frames 400 --cpu-frames 0 --no-prefetch > gpurun_out/${TAG}_c4_noprefetch.json 2> gpurun_out/${TAG}_c4_noprefetch.err
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_c4.json gpurun_out/${TAG}_c4_noprefetch.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","ms_per_frame","q_pipe","ms_per_frame_by_stage","parity_vs_cpu","accepted","max_drift_m")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
( timeout 900 python -m pytest tests/test_mapping_sequence.py tests/test_keyframes.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
