#!/bin/bash
# GPU box: A/B of the E-step's warm-start policy on the driver's command (same box, alternating)
mkdir -p gpurun_out
for rep in 1 2; do for ws in local best; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --warm-start $ws > gpurun_out/ab_${ws}_$rep.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${ws}_$rep.json").read().strip().splitlines()[-1])
print("$ws $rep: %.2f ms/step (E %.2f + M %.2f) cold %.0f; coarse launches %d" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"], d["cold_first_iteration_ms"], d["kernels"]["coarse"]["launches"]))
PY
done; done
