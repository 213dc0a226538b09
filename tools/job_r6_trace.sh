#!/bin/bash
# GPU box: per-round trace of the warm solve of the chr1-sized block, product library and the filter-tally build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 > gpurun_out/r6_trace.out 2> gpurun_out/r6_trace.err || exit 1
grep -A40 -- "---- warm" gpurun_out/r6_trace.err | cut -c1-400; tail -4 gpurun_out/r6_trace.out | cut -c1-600
PHMRF_LIB=variants/libphmrf_fstat.so PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 > gpurun_out/r6_trace_fstat.out 2> gpurun_out/r6_trace_fstat.err || exit 1
echo "---- filter tallies build"; tail -3 gpurun_out/r6_trace_fstat.out | cut -c1-600
