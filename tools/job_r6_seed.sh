#!/bin/bash
# GPU box: the new A/B test, the warm-solve trace, then the strip tests and the driver's command
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_estep.py -x -q -m gpu -k "seed_masks or deterministic or strip or tiles" > gpurun_out/r6_seed_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r6_seed_tests.log
[ $rc -ne 0 ] && exit $rc
export PHMRF_TRACE_PERT=0.05
PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 > gpurun_out/r6_trace2.out 2> gpurun_out/r6_trace2.err || exit 1
grep -A40 -- "---- warm" gpurun_out/r6_trace2.err | grep -v "ms:" | cut -c1-200; tail -4 gpurun_out/r6_trace2.out | cut -c1-700
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit > gpurun_out/r6_seed_bench.json 2> gpurun_out/r6_seed_bench.err || exit 1
python3 -c "
import json;d=json.loads(open('gpurun_out/r6_seed_bench.json').read().strip().splitlines()[-1]);r=d['roofline']
print('bench: ms/step %.2f estep %.2f mstep %.2f cold %.0f | frac %.4f full %s mop %s'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms'],d['cold_first_iteration_ms'],r['frac'],r['full_sweep']['own_ms'],r['mop_up']['own_ms']))"
