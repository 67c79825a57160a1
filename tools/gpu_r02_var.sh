#!/bin/bash
# A/B of solver build variants (instrumented libraries loam_livox_amd/libll_var_*.so)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for lib in loam_livox_amd/libll_var_*.so; do
  name=$(basename $lib .so)
  LOAM_LIVOX_LIB=$GRAFT_REPO_ROOT/$lib timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-q-pipe > gpurun_out/var_$name.json 2> gpurun_out/var_$name.err
  python - "gpurun_out/var_$name.json" "$name" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
    print(sys.argv[2], d["value"], d["kernel_ms_per_step"]["reg_solve_kernel"], d["solver_phase_cycles_scan0"])
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
