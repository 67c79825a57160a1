#!/bin/bash
# phase timers only (timing variant of the library).  usage: bash tools/gpu_timing2.sh <tag> [bench args]
TAG=${1:-x}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe --no-streamed "$@" > gpurun_out/${TAG}_timing.json 2> gpurun_out/${TAG}_timing.err
python - gpurun_out/${TAG}_timing.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    names=["eval","lm","l1","dedupe","select","total","census","prune","table/epi","cnt","c.loadwait","c.insert","c.sums","t.compact","t.planes","t.ids+fill"]
    for key in ("solver_phase_cycles_scan0","single_scan_solver_phase_cycles"):
        v=d[key]; print(key, {n:int(x/10) for n,x in zip(names,v)})
    print({k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","single_scan_latency_ms")}, d.get("roofline",{}).get("avg_launch_ms"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace('.json','.err')).read()[-2000:])
PY
