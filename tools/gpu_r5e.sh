#!/bin/bash
# round 5, call e: small solver with W = 2 / longest-first order: tests, Q-pipe at B = 256 / 2048 (with and without the order), phase cycles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_small.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r5e_tests.log 2>&1
tail -5 gpurun_out/r5e_tests.log
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 4 --warmup 1 --batch 2048 --distinct-scans 256 --q-pipe-in-flight 3 > gpurun_out/r5e_qpipe_b2048.json 2> gpurun_out/r5e_qpipe_b2048.err
LL_DEBUG_OR=262144 timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 4 --warmup 1 --batch 2048 --distinct-scans 256 > gpurun_out/r5e_qpipe_b2048_noorder.json 2> gpurun_out/r5e_qpipe_b2048_noorder.err
LL_DEBUG_OR=65536 timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 4 --warmup 1 --batch 2048 --distinct-scans 256 > gpurun_out/r5e_qpipe_b2048_w1.json 2> gpurun_out/r5e_qpipe_b2048_w1.err
T=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
LOAM_LIVOX_LIB=$T timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 3 --warmup 1 --batch 2048 --distinct-scans 256 > gpurun_out/r5e_timing_b2048.json 2> gpurun_out/r5e_timing_b2048.err
python - <<'PY'
import json
for f in ("r5e_qpipe_b2048","r5e_qpipe_b2048_noorder","r5e_qpipe_b2048_w1","r5e_timing_b2048"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-800:]); continue
    print(f, d["value"], (d.get("sequential") or {}).get("value"), d["kernel_ms_per_step"])
    for k in ("solver_phase_cycles_mean_over_scans","solver_phase_cycles_of_the_slowest_scan"):
        if d.get(k): print("  ",k,d.get(k))
PY
