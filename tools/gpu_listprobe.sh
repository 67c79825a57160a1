#!/bin/bash
# per-call durations of the late-iteration k-NN kernels (kernel trace, in launch order) and the work-list sizes per ICP iteration
# usage: bash tools/gpu_listprobe.sh <tag> [lib]
TAG=${1:-x}; LIBF=$2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -n "$LIBF" ] && export LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/$LIBF
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > /tmp/prof_$TAG/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
python - "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" <<'PY' > gpurun_out/${TAG}_listprobe.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if ('reg_list_kernel' in r['Kernel_Name'] or 'reg_requery' in r['Kernel_Name'] or 'reg_solve' in r['Kernel_Name'] or 'reg_knn_kernel' in r['Kernel_Name']) and int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',0))>10000]
out=[]
for r in sel:
    n=r['Kernel_Name'].split('(')[0].split('::')[-1][:18]
    out.append((n, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0))
# last step only: take the last 10 solves worth
last=[]; cnt=0
for n,d in reversed(out):
    last.append((n,d))
    if n.startswith('reg_solve'): cnt+=1
    if cnt==10 and n.startswith('reg_knn_kernel'): break
for n,d in reversed(last): print(f"{n:20s} {d:8.1f}")
PY
cat gpurun_out/${TAG}_listprobe.txt | awk '{printf "%s:%s  ", $1, $2} END{print ""}'
for it in 3 4 6 10; do
  timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-q-pipe --no-streamed --icp-iters $it 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print($it, d['knn_reuse_last_iter'])"
done
