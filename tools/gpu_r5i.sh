#!/bin/bash
# round 5, call i: key-frame memory rework + ll_cellmap_reserve + cell-map gather: tests; C4 with / without the cell maps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_keyframes.py tests/test_cellmap.py tests/test_ref_cells.py tests/test_mapping_sequence.py tests/test_gpu_multigpu.py -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r5i_tests.log 2>&1
tail -8 gpurun_out/r5i_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5i_c4_400.json 2> gpurun_out/r5i_c4_400.err
timeout 600 python bench_c4.py --frames 400 --no-cell-maps --cpu-frames 0 > gpurun_out/r5i_c4_400_nocells.json 2> gpurun_out/r5i_c4_400_nocells.err
timeout 900 python bench_c4.py --frames 2000 --distinct-frames 200 --cpu-frames 0 > gpurun_out/r5i_c4_2000.json 2> gpurun_out/r5i_c4_2000.err
for f in r5i_c4_400 r5i_c4_400_nocells r5i_c4_2000; do echo == $f; python - gpurun_out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print({k:d.get(k) for k in ("value","ms_per_frame","submap_points_per_rank","submap","gather_s","ms_per_frame_by_stage","parity_vs_cpu","final_drift_m")})
except Exception as e:
    print("failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
