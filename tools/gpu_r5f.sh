#!/bin/bash
# round 5, call f: small solver with the key exchange (no staging array) and the occupancy rule: tests, Q-pipe at B = 2048 / 256
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_small.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r5f_tests.log 2>&1
tail -5 gpurun_out/r5f_tests.log
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 4 --warmup 1 --batch 2048 --distinct-scans 256 --q-pipe-in-flight 3 > gpurun_out/r5f_qpipe_b2048.json 2> gpurun_out/r5f_qpipe_b2048.err
timeout 900 python bench.py --q-pipe --no-cpu-baseline --no-streamed --steps 4 --warmup 1 --batch 1024 --distinct-scans 256 --q-pipe-in-flight 3 > gpurun_out/r5f_qpipe_b1024.json 2> gpurun_out/r5f_qpipe_b1024.err
python - <<'PY'
import json
for f in ("r5f_qpipe_b2048","r5f_qpipe_b1024"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-800:]); continue
    print(f, d["value"], (d.get("sequential") or {}).get("value"), d["kernel_ms_per_step"], d["single_scan_latency_ms"])
PY
