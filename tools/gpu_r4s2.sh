#!/bin/bash
# Q-pipe: batches in flight 3 / 5 / 6.  usage: bash tools/gpu_r4s2.sh <tag>
TAG=${1:-r4s2}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 3 5 6; do
timeout 200 python bench.py --q-pipe --steps 24 --warmup 2 --no-cpu-baseline --no-streamed --q-pipe-in-flight $d > gpurun_out/${TAG}_qpipe_d$d.json 2> gpurun_out/${TAG}_qpipe_d$d.err
python - gpurun_out/${TAG}_qpipe_d$d.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[1], d["value"], d["sequential"]["value"], d["pipeline"]["batches_in_flight"])
except Exception as e:
    print("ERR", sys.argv[1], e)
PY
done
