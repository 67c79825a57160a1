#!/bin/bash
# round 6: the bench lines quoted in DESIGN.md / README.md on the final build: default line, C3, C4 (2000 frames), C5.  usage: bash tools/gpu_r6_final_bench.sh <tag>
TAG=${1:-r06z}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1200 python bench.py ) > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; tail -3 gpurun_out/${TAG}_bench_default.err
timeout 900 python bench_c3.py --cpu-scans 1 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err
timeout 1500 python bench_c4.py --frames 2000 --cpu-frames 8 > gpurun_out/${TAG}_bench_c4_2000frames.json 2> gpurun_out/${TAG}_bench_c4.err
timeout 900 python bench_c5.py > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/${TAG}_bench_c5.err
python - $TAG <<'PY'
import json,sys,os
t=sys.argv[1]
def last(p):
    try: return json.loads(open(p).read().strip().split('\n')[-1])
    except Exception as e: return {"error": repr(e)}
d=last(f"gpurun_out/{t}_bench_default.json")
print("default", {k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms","kernel_ms_per_step")}, (d.get("sequential") or {}).get("value"), (d.get("streamed") or {}).get("value"))
q=d.get("q_pipe") or {}
print("q_pipe", q.get("scans_per_s_this_rank"), q.get("one_batch_at_a_time"), (q.get("at_the_headline_batch_size") or {}).get("scans_per_s_this_rank"), q.get("parity_audit_vs_oracle"))
print("roofline", {k:(d.get("roofline") or {}).get(k) for k in ("frac","avg_launch_ms","traffic","traffic_is_current")})
c=last(f"gpurun_out/{t}_bench_c3.json"); print("c3", {k:c.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","parity_vs_cpu")}, {k:(c.get("roofline") or {}).get(k) for k in ("frac","avg_launch_ms","traffic_over_algorithmic","traffic_is_current")})
c=last(f"gpurun_out/{t}_bench_c4_2000frames.json"); print("c4", {k:c.get(k) for k in ("value","ms_per_frame","frames","error")})
c=last(f"gpurun_out/{t}_bench_c5.json"); print("c5", c.get("value"), {k:(v.get("queries_per_s"), (v.get("roofline") or {}).get("traffic_over_algorithmic")) for k,v in (c.get("runs") or {}).items()})
PY
