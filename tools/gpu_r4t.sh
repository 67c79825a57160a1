#!/bin/bash
# distribution of the solver's per-scan time over a Q-pipe batch (timing library).  usage: bash tools/gpu_r4t.sh <tag>
TAG=${1:-r4t}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 600 python bench.py --q-pipe --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline > gpurun_out/${TAG}_qdist.json 2> gpurun_out/${TAG}_qdist.err
LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-pipeline --no-q-pipe > gpurun_out/${TAG}_fdist.json 2> gpurun_out/${TAG}_fdist.err
python - gpurun_out/${TAG}_qdist.json gpurun_out/${TAG}_fdist.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","kernel_ms_per_step","solver_cycles_per_registration_quantiles","lm_iters_per_scan")})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
