#!/bin/bash
# round 5, call j: cell-map feeder thread + segmented query sort: tests; C4 at 400 / 2000 frames; C3 with the tile search
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests/test_mapping_sequence.py tests/test_cellmap.py "tests/test_gpu_c3_c5.py::test_c3_mid100_deblur_20m_map_matches_oracle" tests/test_gpu_reg.py -m gpu -q 2>&1 | tail -25 ) > gpurun_out/r5j_tests.log 2>&1
tail -8 gpurun_out/r5j_tests.log
timeout 600 python bench_c4.py --frames 400 > gpurun_out/r5j_c4_400.json 2> gpurun_out/r5j_c4_400.err
timeout 900 python bench_c4.py --frames 2000 --distinct-frames 200 --cpu-frames 0 > gpurun_out/r5j_c4_2000.json 2> gpurun_out/r5j_c4_2000.err
for f in r5j_c4_400 r5j_c4_2000; do echo == $f; python - gpurun_out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print({k:d.get(k) for k in ("value","ms_per_frame","submap_points_per_rank","submap","gather_s","ms_per_frame_by_stage","parity_vs_cpu","final_drift_m")})
except Exception as e:
    print("failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
timeout 900 python bench_c3.py --cpu-scans 0 > gpurun_out/r5j_c3.json 2> gpurun_out/r5j_c3.err
LL_DEBUG_OR=1024 timeout 900 python bench_c3.py --cpu-scans 0 > gpurun_out/r5j_c3_tile_reuse.json 2> gpurun_out/r5j_c3_tile_reuse.err
LL_DEBUG_OR=512 timeout 900 python bench_c3.py --cpu-scans 0 > gpurun_out/r5j_c3_notile.json 2> gpurun_out/r5j_c3_notile.err
for f in r5j_c3 r5j_c3_tile_reuse r5j_c3_notile; do echo == $f; python - gpurun_out/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print({k:d.get(k) for k in ("value","ms_per_step","one_batch_at_a_time","kernel_ms_per_step","median_err_vs_truth_m")})
except Exception as e:
    print("failed", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
