#!/bin/bash
# SQ issue counters per kernel (one block at a time so that a kernel's waves are alone on the GPU).  On the GPU box:
#   bash profiles/run_sq_counters.sh   ->  gpurun_out/r1_sq_issue.json   (copy into profiles/)
# (A second pass with the LDS counters SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE did not return within 20 minutes on
#  this pool and is not part of the script; keep the inner timeout.)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/sq1 -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --block-threads 1 > /dev/null 2> gpurun_out/sq1.err
python3 profiles/aggregate_sq.py gpurun_out/sq1 gpurun_out/r1_sq_issue.json
rm -rf gpurun_out/sq1
