#!/bin/bash
# tile tests + default bench + kernel trace + counter passes.  usage: bash tools/gpu_r4d.sh <tag> [pytest -k expression]
TAG=${1:-r4d}; K=${2:-"tile or knn or wavefront"}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py -m gpu -x -q -k "$K" 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 300 python bench.py $C > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms")})
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > /tmp/prof_$TAG/bench.json 2> /tmp/prof_$TAG/trace.log
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_$TAG/trace -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/${TAG}_kernel_trace_by_grid.csv
head -8 gpurun_out/${TAG}_kernel_trace_by_grid.csv | cut -c1-200
bash tools/gpu_pmc3.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
python - <<PY
import csv
rows=list(csv.reader(l for l in open('gpurun_out/${TAG}_pmc_mix.csv') if not l.startswith('#')))
h=rows[0]
for r in rows[1:]:
    if 'tile' in r[0] and int(r[1])>100000:
        print(r[0], r[1], {k:v for k,v in zip(h[3:], r[3:]) if k.replace('_avg','') in ('SQ_INSTS_VALU','SQ_WAVES','SQ_INSTS_LDS','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_INSTS_SALU','SQ_INSTS_VMEM_RD','SQ_BUSY_CYCLES')})
PY
