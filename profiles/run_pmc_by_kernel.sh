#!/bin/bash
# HBM traffic and vector-pipe use per kernel name of the benched build (GPU box): three rocprofv3 --pmc passes of bench.py, blocks one at a time
# and the M-step in-process (no fork under the profiler's preloaded runtime; multi-threaded --pmc passes have hung).
# usage: bash profiles/run_pmc_by_kernel.sh [GIT_REV]  ->  profiles/pmc_by_kernel.json (merged back via gpurun_out/)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
export PHMRF_GIT_REV=${1:-unknown}
O=gpurun_out
CMD="python3 bench.py --steps 2 --warmup 5 --no-cpu-baseline --no-fit --no-through-fit --no-old-tolerance --block-threads 1 --mstep-workers 1"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pk_fetch -- $CMD > $O/pk_fetch.out 2> $O/pk_fetch.err || echo "FETCH pass failed"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pk_write -- $CMD > $O/pk_write.out 2> $O/pk_write.err || echo "WRITE pass failed"
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pk_valu -- $CMD > $O/pk_valu.out 2> $O/pk_valu.err || echo "VALU pass failed"
python3 profiles/aggregate_pmc_by_kernel.py $O/pk_fetch $O/pk_write $O/pmc_by_kernel.json cfg3 "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES (three passes) -- $CMD" $O/pk_valu
rm -rf $O/pk_fetch $O/pk_write $O/pk_valu
