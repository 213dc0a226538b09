cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash profiles/warm_solve_quick.sh r3_base > gpurun_out/r3_base_quick.txt 2>&1
export PHMRF_TRACE_PERT=0.05
PHMRF_LIB=variants/libphmrf_phase.so PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 > gpurun_out/r3_phase.out 2> gpurun_out/r3_phase.err
tail -3 gpurun_out/r3_phase.out
cat gpurun_out/r3_base_quick.txt
