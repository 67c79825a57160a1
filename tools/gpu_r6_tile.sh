#!/bin/bash
# round 6: the tile search: tests that exercise it, the default line, and the phase clocks of the -DLL_TILE_TIMING library.  usage: bash tools/gpu_r6_tile.sh <tag>
TAG=${1:-r06t}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x -k "tile or measured or knn or reg_batch or c3 or mapping" 2>&1 | tail -8 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --no-q-pipe --no-streamed --cpu-runs 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -2 gpurun_out/${TAG}_bench.err
if [ -f loam_livox_amd/libloamlivox_hip_tiletiming.so ]; then
LOAM_LIVOX_LIB=$PWD/loam_livox_amd/libloamlivox_hip_tiletiming.so timeout 900 python bench.py --steps 4 --warmup 1 --no-q-pipe --no-streamed --no-cpu-baseline --no-pipeline > gpurun_out/${TAG}_bench_tile_timing.json 2> gpurun_out/${TAG}_bench_tile_timing.err
tail -3 gpurun_out/${TAG}_bench_tile_timing.err
fi
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench_tile_timing.json <<'PY'
import json,sys,os
names=["tile_query","round_setup","staging","offers","winners_finish","query_pos","sum_T","rounds","store","append","flag","wave_total","waves","waves_listed","listed_lanes","wider_after_stage1"]
for p in sys.argv[1:]:
    if not os.path.exists(p): continue
    d=json.loads(open(p).read().strip().split('\n')[-1])
    print(p, {k:d.get(k) for k in ("value","ms_per_step","sequential","single_scan_latency_ms","kernel_ms_per_step")})
    for k in ("solver_phase_cycles_mean_over_scans","solver_phase_cycles_of_the_slowest_scan"):
        v=d.get(k)
        if v and "timing" in p: print(k, dict(zip(names,v)))
PY
bash tools/gpu_r6_trace.sh ${TAG}
