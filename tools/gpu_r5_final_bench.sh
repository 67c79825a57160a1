#!/bin/bash
# round 5's committed bench lines of the other configurations on the final build: C4 (2000 frames, 500 through the oracle), C3, C5, and the
# Q-pipe figure at 8192 scans per batch.  usage: bash tools/gpu_r5_final_bench.sh <tag>
TAG=${1:-r05a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python bench_c4.py --frames 2000 --cpu-frames 500 > gpurun_out/${TAG}_bench_c4_2000frames.json 2> gpurun_out/${TAG}_bench_c4_2000frames.err
timeout 600 python bench_c3.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
timeout 900 python bench_c5.py > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 3 --warmup 1 --batch 8192 --distinct-scans 256 > gpurun_out/${TAG}_bench_qpipe_b8192.json 2> gpurun_out/${TAG}_bench_qpipe_b8192.err
for f in bench_c4_2000frames bench_c3 bench_c5 bench_qpipe_b8192; do echo "== $f"; tail -c 2500 gpurun_out/${TAG}_$f.json; echo; tail -2 gpurun_out/${TAG}_$f.err | cut -c1-300; done
# hardware-queue experiment (HISTORY.md round 5): the ROCm runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); batches in
# flight sit on their own streams
for Q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$Q timeout 600 python bench.py --no-cpu-baseline --no-streamed --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_hwq$Q.json 2> gpurun_out/${TAG}_bench_hwq$Q.err
  python - gpurun_out/${TAG}_bench_hwq$Q.json $Q <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
q=d.get("q_pipe") or {}
print("GPU_MAX_HW_QUEUES", sys.argv[2], "value", d["value"], "sequential", (d.get("sequential") or {}).get("value"), "q_pipe", q.get("scans_per_s_this_rank"), (q.get("one_batch_at_a_time") or {}).get("scans_per_s_this_rank"))
PY
done
