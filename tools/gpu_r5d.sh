#!/bin/bash
# round 5, call d: phase cycles of the small solver (timing build) + kernel traces of the Q-pipe step at B = 256 / 2048 and of C4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
LOAM_LIVOX_LIB=$T timeout 600 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 3 --warmup 1 > gpurun_out/r5d_timing_b256.json 2> gpurun_out/r5d_timing_b256.err
LOAM_LIVOX_LIB=$T timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --no-pipeline --steps 3 --warmup 1 --batch 2048 --distinct-scans 256 > gpurun_out/r5d_timing_b2048.json 2> gpurun_out/r5d_timing_b2048.err
export TMPDIR=/tmp; mkdir -p /tmp/prof_r5d; cd /tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5d/q -- python $GRAFT_REPO_ROOT/bench.py --q-pipe --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline --batch 2048 --distinct-scans 256 > /tmp/prof_r5d/q.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5d/q256 -- python $GRAFT_REPO_ROOT/bench.py --q-pipe --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline > /tmp/prof_r5d/q256.log 2>&1
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5d/c4 -- python $GRAFT_REPO_ROOT/bench_c4.py --frames 200 --cpu-frames 0 > /tmp/prof_r5d/c4.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_r5d/q -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/r5d_qpipe_b2048_kernels_by_grid.csv
python tools/summarize_rocprof.py trace "$(find /tmp/prof_r5d/q256 -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/r5d_qpipe_b256_kernels_by_grid.csv
python tools/summarize_rocprof.py trace "$(find /tmp/prof_r5d/c4 -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/r5d_c4_kernels_by_grid.csv
tail -2 /tmp/prof_r5d/c4.log | cut -c1-400
python - <<'PY'
import json
for f in ("r5d_timing_b256","r5d_timing_b2048"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-800:]); continue
    print(f, d["value"], d["kernel_ms_per_step"])
    for k in ("solver_phase_cycles_mean_over_scans","solver_phase_cycles_of_the_slowest_scan","solver_cycles_per_registration_quantiles","single_scan_solver_phase_cycles","single_scan_latency_ms"):
        print("  ",k,d.get(k))
PY
for f in r5d_qpipe_b2048 r5d_qpipe_b256 r5d_c4; do echo == $f; head -22 gpurun_out/${f}_kernels_by_grid.csv | cut -d, -f1-3,4,6,8,9,11-15; done
