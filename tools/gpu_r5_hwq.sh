#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for Q in 6 8 12 16 32 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --no-cpu-baseline --no-streamed --steps 20 --warmup 5 > gpurun_out/r5q_hwq$Q.json 2> gpurun_out/r5q_hwq$Q.err
  python - gpurun_out/r5q_hwq$Q.json $Q <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
q=d.get("q_pipe") or {}
print("GPU_MAX_HW_QUEUES", sys.argv[2], "value", d["value"], "sequential", (d.get("sequential") or {}).get("value"), "q_pipe", q.get("scans_per_s_this_rank"), (q.get("one_batch_at_a_time") or {}).get("scans_per_s_this_rank"), "lat", d["single_scan_latency_ms"])
PY
done
