#!/bin/bash
# round 6: C3 (Mid-100 / 20 M / deblur): the product library's line, and the phase cycles of the -DLL_SOLVE_TIMING build.  usage: bash tools/gpu_r6_c3.sh <tag>
TAG=${1:-r06a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python bench_c3.py --cpu-scans 1 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; tail -2 gpurun_out/${TAG}_bench_c3.err
if [ -f loam_livox_amd/libloamlivox_hip_timing.so ]; then
  LOAM_LIVOX_LIB=$PWD/loam_livox_amd/libloamlivox_hip_timing.so timeout 900 python bench_c3.py --cpu-scans 0 --in-flight 1 > gpurun_out/${TAG}_bench_c3_timing.json 2> gpurun_out/${TAG}_bench_c3_timing.err
fi
python - gpurun_out/${TAG}_bench_c3.json gpurun_out/${TAG}_bench_c3_timing.json <<'PY'
import json,sys,os
for p in sys.argv[1:]:
    if not os.path.exists(p): continue
    d=json.loads(open(p).read().strip().split('\n')[-1])
    print(p, {k:d.get(k) for k in ("value","ms_per_step","one_batch_at_a_time","kernel_ms_per_step","blocks_last","parity_vs_cpu","solver_phase_cycles_mean_over_scans","solver_phase_cycles_of_the_slowest_scan")})
    print(" roofline", {k:d["roofline"].get(k) for k in ("frac","avg_launch_ms","traffic_over_algorithmic")})
PY
