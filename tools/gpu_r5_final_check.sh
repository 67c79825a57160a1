#!/bin/bash
# the default bench line at the driver's settings + the whole GPU tier + smoke, on the final build.  usage: bash tools/gpu_r5_final_check.sh <tag>
TAG=${1:-r05a}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 1200 python bench.py ) > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -3 gpurun_out/${TAG}_bench_default.err
python - gpurun_out/${TAG}_bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print({k:d.get(k) for k in ("value","steps","warmup","ms_per_step","sequential","pipeline","streamed","roofline","single_scan_latency_ms","cpu_baseline","parity_vs_cpu")})
print("q_pipe", d.get("q_pipe"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
# the pipelined default loop's timeline (kernel trace only)
export TMPDIR=/tmp; rm -rf /tmp/ptrace_$TAG; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptrace_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-streamed --no-q-pipe > /tmp/ptrace_$TAG.log 2>&1 )
python tools/trace_overlap.py "$(find /tmp/ptrace_$TAG -name '*kernel_trace.csv' | head -1)" 50 > gpurun_out/${TAG}_pipelined_timeline.txt
tail -1 /tmp/ptrace_$TAG.log | cut -c1-300 >> gpurun_out/${TAG}_pipelined_timeline.txt; head -4 gpurun_out/${TAG}_pipelined_timeline.txt
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/${TAG}_tests.log 2>&1
tail -6 gpurun_out/${TAG}_tests.log
