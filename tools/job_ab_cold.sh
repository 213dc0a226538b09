#!/bin/bash
# GPU box: cold solves and the cold first EM iteration per library (variants/libphmrf_NAME.so; "product" = the tree's),
# in turn on ONE box: bash tools/job_ab_cold.sh "base product" [blocks] [bench seed]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
LIBS=${1:-"base product"}; BLOCKS=${2:-"0 9"}; SEED=${3:-0}
for lib in $LIBS $LIBS; do
  if [ $lib = product ]; then unset PHMRF_LIB; else export PHMRF_LIB=$PWD/variants/libphmrf_$lib.so; fi
  echo "== $lib"
  for b in $BLOCKS; do COLD_TWICE=1 python3 tools/cold_trace.py cfg3 $b 2>&1 | grep -v "^{\|amdgpu.ids" ; done
  python3 bench.py --steps 6 --warmup 1 --seed $SEED --no-cpu-baseline --no-fit > gpurun_out/abc_$lib.json 2> gpurun_out/abc_$lib.err
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/abc_$lib.json").read().strip().splitlines()[-1])
print("$lib", "cold first iteration ms", round(d["cold_first_iteration_ms"], 1), "ms/step", round(d["ms_per_step"], 1), "estep by step", [round(x, 1) for x in d["estep_ms_by_step"]])
PY
done
