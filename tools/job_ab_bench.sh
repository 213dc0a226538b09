#!/bin/bash
# GPU box: the driver's command per library (variants/libphmrf_NAME.so; "product" = the tree's), alternating on ONE box:
# bash tools/job_ab_bench.sh "noxcd product" [runs] [extra bench flags]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
LIBS=${1:-"base product"}; RUNS=${2:-2}; EXTRA=$3
for i in $(seq 1 $RUNS); do
  for lib in $LIBS; do
    if [ $lib = product ]; then unset PHMRF_LIB; else export PHMRF_LIB=$PWD/variants/libphmrf_$lib.so; fi
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit $EXTRA > gpurun_out/abb_${lib}_$i.json 2> gpurun_out/abb_${lib}_$i.err
    python3 -c "
import json
d=json.loads(open('gpurun_out/abb_${lib}_$i.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$lib run $i: ms/step %.1f (E %.1f) median %.1f cold %.0f | strip own: full %.0f us mop-up %.0f us frac %.4f | E-step by step %s' % (d['ms_per_step'], d['estep_ms'], d['ms_per_step_median'], d['cold_first_iteration_ms'], r['full_sweep']['avg_launch_us'], r['mop_up']['avg_launch_us'], r['frac'], [round(x) for x in d['estep_ms_by_step']]))"
  done
done
