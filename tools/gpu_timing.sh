#!/bin/bash
# in-kernel phase timers (timing variant of the library), batch and single-scan.  usage: bash tools/gpu_timing.sh <tag> [bench args]
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed "$@" > gpurun_out/${TAG}_timing.json 2> gpurun_out/${TAG}_timing.err
python - gpurun_out/${TAG}_timing.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d[k] for k in ("value","single_scan_latency_ms","solver_phase_cycles_scan0","single_scan_solver_phase_cycles","single_scan_solver")})
PY
