#!/bin/bash
# kernel trace of the C4 mapping loop (per (kernel, grid) durations).  usage: bash tools/gpu_c4trace.sh <tag> [frames]
TAG=${1:-x}; FR=${2:-60}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench_c4.py --frames $FR --cpu-frames 0 > /tmp/prof_$TAG/trace.log 2>&1
tail -2 /tmp/prof_$TAG/trace.log | cut -c1-600
cd "$GRAFT_REPO_ROOT"
F=$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)
python tools/summarize_rocprof.py trace "$F" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_c4_kernel_trace_by_grid.csv
python - gpurun_out/${TAG}_c4_kernel_trace_by_grid.csv $FR <<'PY'
import csv,sys
rows=list(csv.reader(open(sys.argv[1]))); fr=int(sys.argv[2])
tot=sum(float(r[-4]) for r in rows[1:]); calls=sum(int(r[-5]) for r in rows[1:])
print(f"kernel time per frame {tot/fr:.3f} ms in {calls/fr:.0f} launches")
agg={}
for r in rows[1:]:
    a=agg.setdefault(r[0],[0,0.0]); a[0]+=int(r[-5]); a[1]+=float(r[-4])
for k,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:22]:
    print(f"{k:36s} calls/frame {c/fr:6.1f}  ms/frame {t/fr:.4f}  avg_us {1e3*t/c:.1f}")
PY
