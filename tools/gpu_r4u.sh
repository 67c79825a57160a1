#!/bin/bash
# the line-search fit on the controller's wavefront: unit test + registration tests, Q-pipe / Q-full / C4 figures.  usage: bash tools/gpu_r4u.sh <tag>
TAG=${1:-r4u}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_golden.py tests/test_ref_golden.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --q-pipe --steps 8 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_qpipe.json 2> gpurun_out/${TAG}_qpipe.err
python - gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_qpipe.json <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","kernel_ms_per_step","q_pipe","accepted_frac","lm_iters_per_scan")}, d["sequential"]["value"], d["pipeline"]["results_equal_sequential_bitwise"])
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-1500:])
PY
