#!/bin/bash
# round 6: the whole GPU tier + smoke + the default bench line.  usage: bash tools/gpu_r6_check.sh <tag> [pytest args]
TAG=${1:-r06a}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 3000 python -m pytest tests -m gpu -q -x "$@" 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -8 gpurun_out/${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1200 python bench.py ) > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -3 gpurun_out/${TAG}_bench_default.err
python - gpurun_out/${TAG}_bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","ms_per_step","sequential","roofline","single_scan_latency_ms","kernel_ms_per_step")})
print("q_pipe", d.get("q_pipe"))
PY
