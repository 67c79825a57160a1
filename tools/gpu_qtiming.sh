#!/bin/bash
# in-kernel phase timers of the solver on voxel-filtered (Q-pipe) scans: a few hundred blocks on one workgroup.  usage: bash tools/gpu_qtiming.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --q-pipe --batch 16 --distinct-scans 16 > gpurun_out/${TAG}_qtiming.json 2> gpurun_out/${TAG}_qtiming.err
python - gpurun_out/${TAG}_qtiming.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms","solver_phase_cycles_scan0","single_scan_solver_phase_cycles","features_per_scan","lm_iters_per_scan")})
PY
