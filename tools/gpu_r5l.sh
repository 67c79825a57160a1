#!/bin/bash
# round 5, call l: hash + radix select in the 4 / 8-wavefront small solver, voxel ladder: tests; C4; phase cycles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_gpu_small.py tests/test_gpu_voxel.py tests/test_mapping_sequence.py tests/test_keyframes.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r5l_tests.log 2>&1
tail -5 gpurun_out/r5l_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5l_c4_400.json 2> gpurun_out/r5l_c4_400.err
T=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
LOAM_LIVOX_LIB=$T timeout 600 python bench_c4.py --frames 200 --cpu-frames 0 > gpurun_out/r5l_c4_timing.json 2> gpurun_out/r5l_c4_timing.err
timeout 600 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 4 --warmup 1 > gpurun_out/r5l_qpipe_b256.json 2> gpurun_out/r5l_qpipe_b256.err
python - <<'PY'
import json
for f in ("r5l_c4_400","r5l_c4_timing","r5l_qpipe_b256"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_frame","ms_per_step","solver_phase_cycles_last_frame","ms_per_frame_by_stage","parity_vs_cpu","kernel_ms_per_step","single_scan_latency_ms")})
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
