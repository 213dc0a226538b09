#!/bin/bash
# GPU box: warm solve of the chr1-sized block per development build (no hash check): bash tools/job_ab_quick.sh name ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
for v in "$@"; do
  if [ "$v" = "product" ]; then export PHMRF_LIB=""; else export PHMRF_LIB="variants/libphmrf_$v.so"; fi
  python3 tools/trace.py 20 4980 1000 > gpurun_out/ab_$v.out 2> gpurun_out/ab_$v.err
  echo "== $v $(grep -E '^warm solve' gpurun_out/ab_$v.out | cut -c13-50) $(grep -E '^timing' gpurun_out/ab_$v.out | grep -o "'strip': ([0-9.]*" ) $(grep -E '^timing' gpurun_out/ab_$v.out | grep -o "'fusion': ([0-9.]*" )"
done
