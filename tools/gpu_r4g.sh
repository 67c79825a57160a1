#!/bin/bash
# persistence tests + key-frame exploration + bench (persist on / off) + timers + tile counters.  usage: bash tools/gpu_r4g.sh <tag>
TAG=${1:-r4g}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_ref_c2.py tests/test_golden.py -m gpu -x -q -k "persistence or tile or matches_oracle or batch_pipeline or c2 or golden or reuse or determinism or duplicate or subsampl" 2>&1 | tail -30 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
timeout 300 python tools/exp_keyframes.py > gpurun_out/${TAG}_keyframes.txt 2>&1; tail -40 gpurun_out/${TAG}_keyframes.txt | cut -c1-400
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 300 python bench.py $C > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
LL_DEBUG_OR=4096 timeout 300 python bench.py $C > gpurun_out/${TAG}_bench_nopersist.json 2> gpurun_out/${TAG}_bench_nopersist.err
for f in bench bench_nopersist; do python - gpurun_out/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms")}, d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
bash tools/gpu_timing.sh $TAG
bash tools/gpu_pmc3.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
python - <<PY
import csv
rows=list(csv.reader(l for l in open('gpurun_out/${TAG}_pmc_mix.csv') if not l.startswith('#')))
h=rows[0]
for r in rows[1:]:
    if ('tile' in r[0] or 'solve' in r[0]) and int(r[1])>100000:
        print(r[0], r[1], {k:v for k,v in zip(h[3:], r[3:]) if k.replace('_avg','') in ('SQ_INSTS_VALU','SQ_WAVES','SQ_INSTS_LDS','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_BUSY_CYCLES')})
PY
