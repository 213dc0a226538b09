#!/bin/bash
# GPU box: is a slow EM trajectory (seed 4) the critical path of its largest block?  The same command with every block in 2 row tiles.
mkdir -p gpurun_out
for seed in 4 0; do
 for parts in 0 2; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --seed $seed --tile-parts $parts > gpurun_out/st_${seed}_$parts.json 2>gpurun_out/st_${seed}_$parts.err || { tail -5 gpurun_out/st_${seed}_$parts.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/st_${seed}_$parts.json").read().strip().splitlines()[-1])
print("seed $seed parts $parts: %.1f ms/step (E %.1f + M %.1f) %.3e cost1 %s" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"], d["value"], d.get("cost1")))
PY
 done
done
