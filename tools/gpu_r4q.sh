#!/bin/bash
# re-sort schedule of the tile search (A/B through LL_DEBUG_OR) + the pipelined streamed figure.  usage: bash tools/gpu_r4q.sh <tag>
TAG=${1:-r4q}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in 0 8192 16384; do
LL_DEBUG_OR=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-streamed --no-q-pipe > gpurun_out/${TAG}_bench_$v.json 2> gpurun_out/${TAG}_bench_$v.err
done
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-q-pipe > gpurun_out/${TAG}_bench_streamed.json 2> gpurun_out/${TAG}_bench_streamed.err
python - gpurun_out/${TAG}_bench_0.json gpurun_out/${TAG}_bench_8192.json gpurun_out/${TAG}_bench_16384.json gpurun_out/${TAG}_bench_streamed.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","streamed")}, d["sequential"]["value"], d["pipeline"]["results_equal_sequential_bitwise"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
