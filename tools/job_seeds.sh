#!/bin/bash
# GPU box: the driver's command over seeds 0 - 5 on one box (how much of a bench line is the EM trajectory's)
mkdir -p gpurun_out
for seed in 0 1 2 3 4 5; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --seed $seed > gpurun_out/seed_$seed.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/seed_$seed.json").read().strip().splitlines()[-1])
f=d["fit"]
print("seed $seed: %.1f ms/step (E %.1f + M %.1f) %.3e | cold %.0f ms | fit %d iterations (%s) %.3e | coarse launches in the window %d" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"], d["value"], d["cold_first_iteration_ms"], f["iterations"], f["stopped_by"][:14], f["value"], d["kernels"]["coarse"]["launches"]))
PY
done
