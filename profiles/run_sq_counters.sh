#!/bin/bash
# SQ issue counters per kernel (one block at a time so that a kernel's waves are alone on the GPU).  On the GPU box:
#   bash profiles/run_sq_counters.sh   ->  gpurun_out/r2_sq_issue_by_kernel.json   (copy into profiles/)
# Two passes (the counters do not fit one): issue / wait shares of the wave cycles; instruction mix.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/sq1 -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --block-threads 1 > /dev/null 2> gpurun_out/sq1.err
python3 profiles/aggregate_sq.py gpurun_out/sq1 gpurun_out/r2_sq_issue_by_kernel.json
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_IFETCH --output-format csv -d gpurun_out/sq2 -- python3 bench.py --workload cfg2 --steps 2 --warmup 1 --no-cpu-baseline --block-threads 1 > /dev/null 2> gpurun_out/sq2.err
python3 profiles/aggregate_sq.py gpurun_out/sq2 gpurun_out/r2_sq_mix_by_kernel.json
rm -rf gpurun_out/sq1 gpurun_out/sq2
