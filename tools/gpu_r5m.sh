#!/bin/bash
# round 5, call m: history concat in one launch + aabb grid: mapping tests; C4 400 / 2000
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_mapping_sequence.py tests/test_gpu_voxel.py tests/test_verbatim_run.py tests/test_ll_node.py tests/test_adapter_cpp.py tests/test_gpu_full.py -m gpu -q 2>&1 | tail -12 ) > gpurun_out/r5m_tests.log 2>&1
tail -5 gpurun_out/r5m_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5m_c4_400.json 2> gpurun_out/r5m_c4_400.err
timeout 900 python bench_c4.py --frames 2000 --distinct-frames 200 --cpu-frames 0 > gpurun_out/r5m_c4_2000.json 2> gpurun_out/r5m_c4_2000.err
python - <<'PY'
import json
for f in ("r5m_c4_400","r5m_c4_2000"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_frame","ms_per_frame_by_stage","parity_vs_cpu","submap_points_per_rank")})
    except Exception as e:
        print(f, "failed", e); print(open(f"gpurun_out/{f}.err").read()[-1500:])
PY
