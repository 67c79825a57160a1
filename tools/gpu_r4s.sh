#!/bin/bash
# Q-pipe: batches in flight 4 / 6 / 8 (bench.py --q-pipe: the whole line in that query mode).  usage: bash tools/gpu_r4s.sh <tag>
TAG=${1:-r4s}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 4 8 12; do
timeout 600 python bench.py --q-pipe --steps 24 --warmup 2 --no-cpu-baseline --no-streamed --q-pipe-in-flight $d > gpurun_out/${TAG}_qpipe_d$d.json 2> gpurun_out/${TAG}_qpipe_d$d.err
done
python - gpurun_out/${TAG}_qpipe_d4.json gpurun_out/${TAG}_qpipe_d8.json gpurun_out/${TAG}_qpipe_d12.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","accepted_frac")}, d["sequential"]["value"], d["pipeline"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
