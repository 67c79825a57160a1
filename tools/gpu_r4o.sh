#!/bin/bash
# depth of the batch pipeline.  usage: bash tools/gpu_r4o.sh <tag>
TAG=${1:-r4o}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for d in 3 4; do
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-streamed --no-q-pipe --in-flight $d > gpurun_out/${TAG}_bench_d$d.json 2> gpurun_out/${TAG}_bench_d$d.err
done
python - gpurun_out/${TAG}_bench_d3.json gpurun_out/${TAG}_bench_d4.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","pipeline")}, d["sequential"]["value"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
