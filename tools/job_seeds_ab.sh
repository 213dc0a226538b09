#!/bin/bash
# GPU box: the driver's command over seeds, product library against variants/libphmrf_base.so, alternating on ONE box:
# bash tools/job_seeds_ab.sh "1 2 3 5"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for s in ${1:-"1 2 3"}; do
  for lib in base product; do
    if [ $lib = product ]; then unset PHMRF_LIB; else export PHMRF_LIB=$PWD/variants/libphmrf_$lib.so; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --seed $s > gpurun_out/sab_${s}_$lib.json 2> gpurun_out/sab_${s}_$lib.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/sab_${s}_$lib.json').read().strip().splitlines()[-1])
print('seed $s $lib: ms/step %.1f (E %.1f) median %.1f cold %.0f | E-step by step %s | cost1 %.3f' % (d['ms_per_step'], d['estep_ms'], d['ms_per_step_median'], d['cold_first_iteration_ms'], [round(x) for x in d['estep_ms_by_step']], d['cost1'][-1]))"
  done
done
