cd "$GRAFT_REPO_ROOT"
export PHMRF_TRACE_PERT=0.05
PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 > gpurun_out/r3_trace.out 2> gpurun_out/r3_trace.err
grep -A30 -- "---- warm" gpurun_out/r3_trace.err | cut -c1-330
