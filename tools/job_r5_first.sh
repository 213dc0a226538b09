#!/bin/bash
# round 5, first GPU call: state dump for the filter probe + baseline warm-solve profile + phase clocks
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python3 tools/dump_state.py 20 700 gpurun_out/state_700.npz > gpurun_out/dump.out 2>&1 && tail -3 gpurun_out/dump.out
bash profiles/warm_solve_quick.sh r5base
export PHMRF_TRACE_PERT=0.05
PHMRF_LIB=variants/libphmrf_phase.so python3 tools/trace.py 20 4980 1000 2> gpurun_out/phase.err | python3 tools/phase_report.py > gpurun_out/phase_report.txt; cat gpurun_out/phase_report.txt | tail -30
