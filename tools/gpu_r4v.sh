#!/bin/bash
# controller with register copies of the evaluation again: Q-full / Q-pipe figures, twice each.  usage: bash tools/gpu_r4v.sh <tag>
TAG=${1:-r4v}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for i in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench$i.json 2> gpurun_out/${TAG}_bench$i.err
done
python - gpurun_out/${TAG}_bench1.json gpurun_out/${TAG}_bench2.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step")}, d["sequential"]["value"], d["q_pipe"]["scans_per_s_this_rank"], d["q_pipe"]["one_batch_at_a_time"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
