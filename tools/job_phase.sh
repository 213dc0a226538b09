cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export PHMRF_TRACE_PERT=0.05
PHMRF_LIB=variants/libphmrf_${1:-phase}.so python3 tools/trace.py 20 4980 1000 > gpurun_out/r3_phase.out 2> gpurun_out/r3_phase.err
tail -2 gpurun_out/r3_phase.out
