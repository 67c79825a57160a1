#!/bin/bash
# solver bring-up / A-B: registration tests under a short timeout, then the bench with the plane-table path and with the
# round-2 packed-record path, then the in-kernel phase timers of both.  usage: bash tools/gpu_ab.sh <tag> [extra bench args for the B leg]
TAG=${1:-x}; shift
BARGS=${@:---packed48-solver}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_golden.py tests/test_ref_golden.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${TAG}_t1.log 2>&1
tail -8 gpurun_out/${TAG}_t1.log
fi
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
C="--steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe --no-streamed"
timeout 600 python bench.py $C > gpurun_out/${TAG}_bench_a.json 2> gpurun_out/${TAG}_bench_a.err
timeout 600 python bench.py $C $BARGS > gpurun_out/${TAG}_bench_b.json 2> gpurun_out/${TAG}_bench_b.err
export LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed > gpurun_out/${TAG}_timing_a.json 2> gpurun_out/${TAG}_timing_a.err
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed $BARGS > gpurun_out/${TAG}_timing_b.json 2> gpurun_out/${TAG}_timing_b.err
for f in bench_a bench_b timing_a timing_b; do python - gpurun_out/${TAG}_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[1].split('_',1)[1], {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms","solver_phase_cycles_scan0","single_scan_solver_phase_cycles")}, d.get("roofline",{}).get("avg_launch_ms"), d.get("parity_vs_cpu"))
except Exception as e:
    print("ERR", sys.argv[1], e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
done
