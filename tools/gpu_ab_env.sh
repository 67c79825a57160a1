#!/bin/bash
# A/B of an environment switch in ONE call: single-scan latency / batch rate / C4 loop with and without it, twice each, interleaved.
# usage: bash tools/gpu_ab_env.sh <tag> <ENV_NAME>
TAG=${1:-x}; EV=$2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
C="--steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed"
for rep in 1 2; do
  timeout 300 python bench.py $C > gpurun_out/${TAG}_a$rep.json 2>/dev/null
  env $EV=1 timeout 300 python bench.py $C > gpurun_out/${TAG}_b$rep.json 2>/dev/null
  timeout 300 python bench_c4.py --frames 300 --cpu-frames 0 > gpurun_out/${TAG}_c4a$rep.json 2>/dev/null
  env $EV=1 timeout 300 python bench_c4.py --frames 300 --cpu-frames 0 > gpurun_out/${TAG}_c4b$rep.json 2>/dev/null
done
python - $TAG <<'PY'
import json,sys
t=sys.argv[1]
for n in ("a1","b1","a2","b2","c4a1","c4b1","c4a2","c4b2"):
    try:
        d=json.loads(open(f"gpurun_out/{t}_{n}.json").read().strip().split('\n')[-1])
        print(n, {k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms","ms_per_frame","ms_per_frame_by_stage") if d.get(k) is not None})
    except Exception as e: print(n, "ERR", e)
PY
