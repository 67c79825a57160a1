#!/bin/bash
# round 6: the headline solver: tests that exercise it, the default line, the phase clocks of the -DLL_SOLVE_TIMING library, kernel trace.  usage: bash tools/gpu_r6_solver.sh <tag>
TAG=${1:-r06s}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x -k "reg or measured or c3" 2>&1 | tail -8 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log
timeout 900 python bench.py --no-q-pipe --no-streamed --cpu-runs 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
if [ -f loam_livox_amd/libloamlivox_hip_timing.so ]; then
LOAM_LIVOX_LIB=$PWD/loam_livox_amd/libloamlivox_hip_timing.so timeout 900 python bench.py --steps 4 --warmup 1 --no-q-pipe --no-streamed --no-cpu-baseline --no-pipeline > gpurun_out/${TAG}_bench_timing.json 2> gpurun_out/${TAG}_bench_timing.err
fi
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_bench_timing.json <<'PY'
import json,sys,os
names=["eval","controller","L1","dedupe","select","total","census","prune","table","x9","census_load_wait","inserts","block_sums","id_compaction","plane_consts","id_pass"]
for p in sys.argv[1:]:
    if not os.path.exists(p): continue
    d=json.loads(open(p).read().strip().split('\n')[-1])
    print(p, {k:d.get(k) for k in ("value","ms_per_step","single_scan_latency_ms","kernel_ms_per_step")}, (d.get("sequential") or {}).get("value"))
    v=d.get("solver_phase_cycles_mean_over_scans")
    if v: print(dict(zip(names,v)))
PY
bash tools/gpu_r6_trace.sh ${TAG} | grep "solve_kernel\|tile_kernel\|lane_kernel"
