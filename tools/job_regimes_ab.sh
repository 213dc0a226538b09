#!/bin/bash
# GPU box: two data regimes of the synthetic generator and seed 4 of the default one (its far-off iteration), the product
# library against variants/libphmrf_base.so, alternating on ONE box: bash tools/job_regimes_ab.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "8 1.0 0" "25 1.6 0" "25 1.0 4"; do
  set -- $spec
  for lib in base product; do
    if [ $lib = product ]; then unset PHMRF_LIB; else export PHMRF_LIB=$PWD/variants/libphmrf_$lib.so; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --mean-run $1 --noise $2 --seed $3 > gpurun_out/rab_$1_$2_$3_$lib.json 2> gpurun_out/rab_$1_$2_$3_$lib.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/rab_$1_$2_$3_$lib.json').read().strip().splitlines()[-1])
print('mean run $1 noise $2 seed $3 $lib: ms/step %.1f (E %.1f) median %.1f cold %.0f | E-step by step %s | cost1 %.3f' % (d['ms_per_step'], d['estep_ms'], d['ms_per_step_median'], d['cold_first_iteration_ms'], [round(x) for x in d['estep_ms_by_step']], d['cost1'][-1]))"
  done
done
