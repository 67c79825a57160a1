#!/bin/bash
# round 5, call k: where a C4 frame goes: solver phase cycles (timing build) and a kernel trace; AB-library tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
LOAM_LIVOX_LIB=$T timeout 600 python bench_c4.py --frames 200 --cpu-frames 0 > gpurun_out/r5k_c4_timing.json 2> gpurun_out/r5k_c4_timing.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5k_c4_timing.json").read().strip().split("\n")[-1])
print({k:d.get(k) for k in ("value","ms_per_frame","solver_phase_cycles_last_frame","lm_iterations_last_frame","blocks_last_frame","ms_per_frame_by_stage")})
PY
export TMPDIR=/tmp; mkdir -p /tmp/prof_r5k; cd /tmp
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r5k/c4 -- python $GRAFT_REPO_ROOT/bench_c4.py --frames 200 --cpu-frames 0 > /tmp/prof_r5k/c4.log 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/summarize_rocprof.py trace "$(find /tmp/prof_r5k/c4 -name '*kernel_trace.csv' | head -1)" loam_livox_amd/libloamlivox_hip.so > gpurun_out/r5k_c4_kernels_by_grid.csv
python - <<'PY'
import csv
rows=list(csv.reader(open('gpurun_out/r5k_c4_kernels_by_grid.csv')))
hdr=rows[0]; n=len(hdr); tot=0
for r in rows[1:]:
    name=','.join(r[:len(r)-(n-1)]); d=dict(zip(hdr[1:], r[len(r)-(n-1):])); tot+=float(d['total_ms'])
print("total kernel ms over 206 frames", round(tot,1))
for r in rows[1:26]:
    name=','.join(r[:len(r)-(n-1)]); d=dict(zip(hdr[1:], r[len(r)-(n-1):]))
    print(f"{name[:60]:60s} grid={d['grid_threads']:>8s} calls={d['calls']:>5s} total={d['total_ms']:>8s} avg_us={d['avg_us']:>7s}")
PY
( timeout 900 python -m pytest tests/test_gpu_reg.py -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r5k_tests.log 2>&1
tail -4 gpurun_out/r5k_tests.log
