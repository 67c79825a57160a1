#!/bin/bash
# round 5, call n: one-workgroup voxel filter for any number of clouds: voxel tests, Q-pipe at 2048 / 256
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_small.py -m gpu -q 2>&1 | tail -8 ) > gpurun_out/r5n_tests.log 2>&1
tail -4 gpurun_out/r5n_tests.log
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 4 --warmup 1 --batch 2048 --distinct-scans 256 --q-pipe-in-flight 3 > gpurun_out/r5n_qpipe_b2048.json 2> gpurun_out/r5n_qpipe_b2048.err
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 6 --warmup 2 > gpurun_out/r5n_qpipe_b256.json 2> gpurun_out/r5n_qpipe_b256.err
python - <<'PY'
import json
for f in ("r5n_qpipe_b2048","r5n_qpipe_b256"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
        print(f, d["value"], (d.get("sequential") or {}).get("value"), d["kernel_ms_per_step"], d["ms_per_step"], d["single_scan_latency_ms"])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
