#!/bin/bash
# grouped-solver exchange bring-up: the tests of the grouped / single-workgroup forms, then the single-scan latency and a batch of 16
# usage: bash tools/gpu_xch.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 700 python -m pytest tests/test_gpu_reg.py tests/test_golden.py -m gpu -x -q -k "grouped or abort or (registration_matches and not legacy) or plane_table or bounded or batch_pipeline or determinism or golden" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 400 python bench.py $C > gpurun_out/${TAG}_bench_a.json 2> gpurun_out/${TAG}_bench_a.err
timeout 400 python bench.py $C --batch 16 --distinct-scans 16 > gpurun_out/${TAG}_bench_16.json 2> gpurun_out/${TAG}_bench_16.err
export LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
[ -f $LOAM_LIVOX_LIB ] && timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > gpurun_out/${TAG}_timing.json 2> gpurun_out/${TAG}_timing.err
for f in bench_a bench_16 timing; do python - gpurun_out/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('_',1)[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms","single_scan_solver_phase_cycles")}, d.get("parity_vs_cpu"))
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
