#!/bin/bash
# the default bench line at the driver's settings + the whole GPU tier + smoke, on the final build.  usage: bash tools/gpu_r4_final_check.sh <tag>
TAG=${1:-r04b}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json; echo
python - gpurun_out/${TAG}_bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","steps","warmup","ms_per_step","sequential","pipeline","streamed","q_pipe","roofline_knn")})
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
