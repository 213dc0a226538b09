#!/bin/bash
# GPU box: what the HIP-event timers around every kernel class cost in the timed region (driver's command, alternating)
mkdir -p gpurun_out
for rep in 1 2; do for t in on off; do
  if [ $t = off ]; then F="--no-kernel-timing"; else F=""; fi
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit $F > gpurun_out/abt_${t}_$rep.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/abt_${t}_$rep.json").read().strip().splitlines()[-1])
print("timers $t $rep: %.2f ms/step (E %.2f + M %.2f)" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"]))
PY
done; done
