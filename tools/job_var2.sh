#!/bin/bash
# GPU box: the warm solve of the chr1-sized block (tools/trace.py) with the product library and with development variants
# usage: bash tools/job_var2.sh name1 name2 ...   (variants/libphmrf_NAME.so; "product" = the product library)
mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
for v in "$@"; do
  if [ "$v" = "product" ]; then LIBV=""; else LIBV="variants/libphmrf_$v.so"; fi
  PHMRF_LIB=$LIBV python3 tools/trace.py 20 4980 1000 > gpurun_out/var_$v.out 2> gpurun_out/var_$v.err
  echo "== $v"; grep -E "warm solve|work:" gpurun_out/var_$v.out | cut -c1-400
  python3 - <<PY
import re
# per-class ms of the warm solve from the timing dict printed by trace.py? (not printed): use PHMRF_SOLVE_TRACE lines if any
PY
done
