#!/bin/bash
# Per-dispatch view of ONE warm-start solve on the chr1 block of the 50 kb workload (tools/trace.py: cold solve, 5 %
# parameter perturbation, warm solve): kernel durations in launch order and SQ counters of the warm part.
# usage (GPU box): bash profiles/warm_solve_profile.sh TAG  ->  gpurun_out/TAG_warm_*.json
TAG=${1:-r2}
N=${2:-4980}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out
export PHMRF_TRACE_PERT=0.05
rocprofv3 --kernel-trace --output-format csv -d $O/ws_kt -- python3 tools/trace.py 20 $N 1000 > $O/${TAG}_warm_trace.out 2> $O/${TAG}_warm_trace.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/ws_pm -- python3 tools/trace.py 20 $N 1000 > /dev/null 2> $O/${TAG}_warm_pmc.err
python3 profiles/warm_solve_aggregate.py $O/ws_kt $O/ws_pm $O/${TAG}_warm_solve.json
rm -rf $O/ws_kt $O/ws_pm
