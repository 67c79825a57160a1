#!/bin/bash
# LDS copy of the solver's line blocks, A/B (LL_DEBUG_OR=4096 switches it off): phase timers on small scans, C4, Q-full bench, then tests.
TAG=${1:-r4m}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TL=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so
show() { python - "$@" <<'PY'
import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().split('\n')[-1])
        print(f, {k:d.get(k) for k in ("value","ms_per_step","ms_per_frame","kernel_ms_per_step","single_scan_latency_ms","solver_phase_cycles_scan0","solver_phase_cycles_last_frame","q_pipe") if d.get(k) is not None})
    except Exception as e:
        print("ERR", f, e); print(open(f.replace('.json','.err')).read()[-800:])
PY
}
for v in on off; do
  if [ $v = off ]; then export LL_DEBUG_OR=4096; else unset LL_DEBUG_OR; fi
  LOAM_LIVOX_LIB=$TL timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-streamed --q-pipe --batch 16 --distinct-scans 16 > gpurun_out/${TAG}_qtiming_$v.json 2> gpurun_out/${TAG}_qtiming_$v.err
  LOAM_LIVOX_LIB=$TL timeout 300 python bench_c4.py --frames 200 --cpu-frames 0 > gpurun_out/${TAG}_c4timing_$v.json 2> gpurun_out/${TAG}_c4timing_$v.err
  timeout 300 python bench_c4.py --frames 400 --cpu-frames 0 > gpurun_out/${TAG}_c4_$v.json 2> gpurun_out/${TAG}_c4_$v.err
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed > gpurun_out/${TAG}_bench_$v.json 2> gpurun_out/${TAG}_bench_$v.err
  show gpurun_out/${TAG}_qtiming_$v.json gpurun_out/${TAG}_c4timing_$v.json gpurun_out/${TAG}_c4_$v.json gpurun_out/${TAG}_bench_$v.json
done
unset LL_DEBUG_OR
( timeout 1200 python -m pytest tests/test_gpu_reg.py tests/test_golden.py tests/test_ref_golden.py tests/test_mapping_sequence.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -5 gpurun_out/${TAG}_tests.log
