#!/bin/bash
# GPU box: A/B of development builds.  usage: bash tools/job_ab.sh name1 name2 ...   ("product" = the product library,
# otherwise variants/libphmrf_NAME.so).  Per build: cold solves of two cfg3 blocks under PHMRF_DETERMINISTIC=1 (labels' SHA-1,
# energy, rounds -- a kernel change that must not move a label shows here), then the warm solve of the chr1-sized block
# (tools/trace.py: device ms per kernel class).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
for v in "$@"; do
  if [ "$v" = "product" ]; then export PHMRF_LIB=""; else export PHMRF_LIB="variants/libphmrf_$v.so"; fi
  echo "== $v"
  for spec in "9 4" "0 0"; do
    set -- $spec
    PHMRF_DETERMINISTIC=1 python3 tools/cold_trace.py cfg3 $1 $2 2>/dev/null | grep -E "sha1|cold solve" | cut -c1-200
  done
  for rep in 1 2; do
    python3 tools/trace.py 20 4980 1000 > gpurun_out/ab_$v.out 2> gpurun_out/ab_$v.err
    grep -E "^warm solve" gpurun_out/ab_$v.out | cut -c1-120
    grep -E "^timing" gpurun_out/ab_$v.out | cut -c1-400
  done
done
