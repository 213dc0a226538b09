#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
for v in "$@"; do
  echo "== $v"
  PHMRF_LIB=variants/libphmrf_$v.so python3 tools/trace.py 20 4980 1000 2> gpurun_out/ph_$v.err | tee gpurun_out/ph_$v.out | python3 tools/phase_report.py
  grep -E "^timing" gpurun_out/ph_$v.out | cut -c1-300
done
