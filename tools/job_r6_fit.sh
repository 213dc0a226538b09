#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_fit.py tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r6_fit_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r6_fit_tests.log
[ $rc -ne 0 ] && exit $rc
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6_fit_bench.json 2> gpurun_out/r6_fit_bench.err || { tail -20 gpurun_out/r6_fit_bench.err; exit 1; }
python3 -c "
import json;d=json.loads(open('gpurun_out/r6_fit_bench.json').read().strip().splitlines()[-1]);r=d['roofline']
print('bench: ms/step %.2f estep %.2f mstep %.2f cold %.0f | frac %.4f full %s mop %s'%(d['ms_per_step'],d['estep_ms'],d['mstep_ms'],d['cold_first_iteration_ms'],r['frac'],r['full_sweep']['own_ms'],r['mop_up']['own_ms']))
print('fit_surface', {k:v for k,v in d['fit_surface'].items() if k!='how'})
print('fit', d['fit']['value'], d['fit']['iterations'])"
