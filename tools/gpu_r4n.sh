#!/bin/bash
# two batches in flight (bench.py default) against one at a time; the A/B library on the device.  usage: bash tools/gpu_r4n.sh <tag>
TAG=${1:-r4n}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streamed --no-q-pipe > gpurun_out/${TAG}_bench20.json 2> gpurun_out/${TAG}_bench20.err
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench20.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","sequential","pipeline","kernel_ms_per_step","streamed","q_pipe")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
( timeout 900 python -m pytest tests/test_gpu_reg.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
