#!/bin/bash
# sequential-path figures: the bench's Q-pipe / streamed / single-scan legs and the C4 mapping loop.  usage: bash tools/gpu_seq.sh <tag> [frames] [cpu frames] [distinct]
TAG=${1:-x}; FR=${2:-300}; CF=${3:-100}; DF=${4:-0}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 1500 python bench_c4.py --frames $FR --cpu-frames $CF --distinct-frames $DF > gpurun_out/${TAG}_c4.json 2> gpurun_out/${TAG}_c4.err
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_c4.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms","ms_per_frame","ms_per_frame_by_stage","frames_per_sequence","parity_vs_cpu","cpu_baseline")})
        for k in ("streamed","q_pipe"):
            if k in d: print("   ", k, {kk:d[k][kk] for kk in d[k] if kk in ("value","ms_per_step","scans_per_s_this_rank")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
