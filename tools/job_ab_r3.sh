#!/bin/bash
# GPU box: the round-3 tree (variants/r3tree, built in the container) against this tree, driver's command, alternating
mkdir -p gpurun_out
for rep in 1 2; do
  (cd variants/r3tree && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > ../../gpurun_out/ab_r3_$rep.json 2>/dev/null)
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --warm-start local > gpurun_out/ab_r4local_$rep.json 2>/dev/null
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit > gpurun_out/ab_r4best_$rep.json 2>/dev/null
  for t in r3 r4local r4best; do python3 - <<PY
import json
d=json.loads(open("gpurun_out/ab_${t}_$rep.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$t $rep: %.2f ms/step (E %.2f + M %.2f); stream ms/step: %s" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"], {n: round(v["ms"]/20,1) for n,v in k.items() if v["launches"]}))
PY
  done
done
