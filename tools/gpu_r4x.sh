#!/bin/bash
# counting sort of the queries by cell: registration tests + bench.  usage: bash tools/gpu_r4x.sh <tag>
TAG=${1:-r4x}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_golden.py tests/test_ref_golden.py tests/test_ref_c2.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -4 gpurun_out/${TAG}_tests.log
timeout 600 python bench.py --no-cpu-baseline --no-streamed --no-q-pipe > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step")}, d["sequential"]["value"], d["pipeline"]["results_equal_sequential_bitwise"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
