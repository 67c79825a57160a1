#!/bin/bash
# round-2 GPU call A: parity of the new solver path + A/B of the solver variants (timed and instrumented)
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_reg.py tests/test_ref_golden.py tests/test_golden.py tests/test_gpu_full.py -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/a_tests.log 2>&1
for v in new legacy; do
  flag=""; [ $v = legacy ] && flag="--legacy-solver"
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-q-pipe $flag > gpurun_out/a_bench_$v.json 2> gpurun_out/a_bench_$v.err
  LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/loam_livox_amd/libloamlivox_hip_timing.so timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe $flag > gpurun_out/a_timing_$v.json 2> gpurun_out/a_timing_$v.err
done
tail -5 gpurun_out/a_tests.log
for f in gpurun_out/a_bench_*.json gpurun_out/a_timing_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print({k:d[k] for k in ("value","ms_per_step","kernel_ms_per_step","solver_phase_cycles_scan0","single_scan_latency_ms")}, d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("ERR", e)
PY
done
# rocprofv3: kernel trace, then HBM byte counters in their own passes (new default solver)
cd /tmp
rm -rf /tmp/prof_a && mkdir -p /tmp/prof_a
B="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a/trace -- $B > /tmp/prof_a/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_a/fetch -- $B > /tmp/prof_a/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_a/write -- $B > /tmp/prof_a/write.log 2>&1
cd "$GRAFT_REPO_ROOT"
T=$(find /tmp/prof_a/trace -name '*kernel_trace.csv' | head -1)
F=$(find /tmp/prof_a/fetch -name '*counter_collection.csv' | head -1)
W=$(find /tmp/prof_a/write -name '*counter_collection.csv' | head -1)
python tools/summarize_rocprof.py trace "$T" > gpurun_out/a_kernel_trace_by_grid.csv
python tools/summarize_rocprof.py pmc "$F" "$W" > gpurun_out/a_pmc_hbm_bytes.csv
cp $(find /tmp/prof_a/trace -name '*kernel_stats.csv' | head -1) gpurun_out/a_kernel_stats_raw.csv
head -12 gpurun_out/a_kernel_trace_by_grid.csv; head -6 gpurun_out/a_pmc_hbm_bytes.csv
