#!/bin/bash
# GPU box: the E-step of every timed step, by seed and number of warm-up steps (is a slow step the trajectory's or the bench's?)
mkdir -p gpurun_out
for cfg in "4 5" "4 6" "4 3" "0 5" "0 6"; do
  set -- $cfg
  python3 bench.py --steps 20 --warmup $2 --no-cpu-baseline --no-fit --seed $1 > gpurun_out/steps_$1_$2.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/steps_$1_$2.json").read().strip().splitlines()[-1])
print("seed $1 warmup $2: %.1f ms/step cold %.0f  E by step %s" % (d["ms_per_step"], d["cold_first_iteration_ms"], d["estep_ms_by_step"]))
PY
done
