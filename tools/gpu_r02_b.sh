#!/bin/bash
# round-2 GPU call B: solver 2b (L1 shortcut, 2-bit de-dup, reciprocal Cholesky) -- parity + timing
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_ref_golden.py tests/test_golden.py tests/test_gpu_full.py tests/test_mapping_sequence.py tests/test_adapter_cpp.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/i_tests.log 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe > gpurun_out/i_bench_new.json 2> gpurun_out/i_bench_new.err
LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe > gpurun_out/i_timing_new.json 2> gpurun_out/i_timing_new.err
tail -5 gpurun_out/i_tests.log
for f in gpurun_out/i_bench_new.json gpurun_out/i_timing_new.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print({k:d[k] for k in ("value","ms_per_step","kernel_ms_per_step","solver_phase_cycles_scan0","single_scan_latency_ms")}, d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
done
