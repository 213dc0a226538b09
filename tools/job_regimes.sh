#!/bin/bash
# GPU box: the driver's command on other DATA REGIMES of the synthetic generator (the label solver's schedule was tuned on
# mean run 25, noise 1): bash tools/job_regimes.sh  -> gpurun_out/regime_*.json
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for spec in "25 1.0" "8 1.0" "60 1.0" "25 0.5" "25 1.6"; do
  set -- $spec
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --mean-run $1 --noise $2 > gpurun_out/regime_$1_$2.json 2> gpurun_out/regime_$1_$2.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/regime_$1_$2.json').read().strip().splitlines()[-1])
f=d['fit']
print('mean run $1 noise $2: ms/step %.1f (E %.1f + M %.1f) median %.1f cold %.0f | fit %d it %.2e | cost1 %.3f' % (d['ms_per_step'], d['estep_ms'], d['mstep_ms'], d['ms_per_step_median'], d['cold_first_iteration_ms'], f['iterations'], f['value'], d['cost1'][-1]))"
done
