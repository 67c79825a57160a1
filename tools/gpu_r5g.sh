#!/bin/bash
# round 5, call g: W = 8 / LDS sort: small-solver tests + mapping tests; C4; default bench line (new Q-pipe figures)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_small.py tests/test_mapping_sequence.py tests/test_gpu_voxel.py tests/test_keyframes.py -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r5g_tests.log 2>&1
tail -5 gpurun_out/r5g_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5g_c4.json 2> gpurun_out/r5g_c4.err
tail -c 900 gpurun_out/r5g_c4.json; echo
( time timeout 1500 python bench.py ) > gpurun_out/r5g_bench_default.json 2> gpurun_out/r5g_bench_default.err
tail -4 gpurun_out/r5g_bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5g_bench_default.json").read().strip().split("\n")[-1])
print({k:d.get(k) for k in ("value","ms_per_step","sequential","q_pipe","single_scan_latency_ms","cpu_baseline","cpu_baseline_shipped_config_allcores","parity_vs_cpu")})
PY
