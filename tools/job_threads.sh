#!/bin/bash
# GPU box: the driver's command with different numbers of blocks in flight, on one box: bash tools/job_threads.sh "10 14 18 22" [runs]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
RUNS=${2:-2}
for i in $(seq 1 $RUNS); do
for t in $1; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --block-threads $t > gpurun_out/thr_${t}_$i.json 2> gpurun_out/thr_${t}_$i.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/thr_${t}_$i.json').read().strip().splitlines()[-1])
print('threads $t run $i: ms/step %.1f E-step %.1f median %.1f' % (d['ms_per_step'], d['estep_ms'], d['ms_per_step_median']))"
done
done
