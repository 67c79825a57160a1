#!/bin/bash
# wavefront-cooperative corner search bring-up / A-B: the k-NN and registration tests, the bench with and without it, and the per-call
# durations of the late-iteration kernels.  usage: bash tools/gpu_coop.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_reg.py -m gpu -x -q -k "knn5 or wavefront or flat or reuse or determinism or (registration_matches and not legacy) or batch_pipeline or grouped_and" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 400 python bench.py $C > gpurun_out/${TAG}_bench_a.json 2> gpurun_out/${TAG}_bench_a.err
timeout 400 python bench.py $C ${BFLAG:---no-knn-coop} > gpurun_out/${TAG}_bench_b.json 2> gpurun_out/${TAG}_bench_b.err
for f in bench_a bench_b; do python - gpurun_out/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('_',1)[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms","knn_reuse_last_iter")}, d.get("parity_vs_cpu"))
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
done
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > /tmp/prof_$TAG/trace.log 2>&1
cd "$GRAFT_REPO_ROOT"
F=$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)
python tools/summarize_rocprof.py trace "$F" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
python - "$F" <<'PY' > gpurun_out/${TAG}_listprobe.txt
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
out=[]
for r in rows:
    n=r['Kernel_Name']
    if not any(k in n for k in ('reg_list_kernel','reg_requery','reg_solve','reg_knn')): continue
    g=int(r['Grid_Size_X'])*int(r['Grid_Size_Y'])*int(r['Grid_Size_Z'])
    out.append((n.split('(')[0].split('::')[-1][:26], g, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000.0))
# the last batch step (large grids) and the last single-scan registration (small grids), in launch order
big=[o for o in out if o[1]>100000 or (o[0].startswith('reg_solve') and o[1]>50000)]
small=[o for o in out if not (o[1]>100000 or (o[0].startswith('reg_solve') and o[1]>50000))]
def last_reg(seq):
    res=[]; cnt=0
    for o in reversed(seq):
        res.append(o)
        if o[0].startswith('reg_solve'): cnt+=1
        if cnt==10 and o[0].startswith('reg_knn_kernel'): break
    return list(reversed(res))
for name,seq in (('batch',big),('single',small)):
    print('#',name)
    for n,g,d in last_reg(seq): print(f"{n:28s} {g:9d} {d:8.1f}")
PY
cat gpurun_out/${TAG}_listprobe.txt | awk '{printf "%s:%s  ", $1, $3} END{print ""}'
head -30 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -d, -f1-3,11-15
