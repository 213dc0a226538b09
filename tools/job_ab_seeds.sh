#!/bin/bash
# GPU box: the round-3 tree against this tree on the driver's command over several seeds (is a difference the build's or the
# EM trajectory's?)
mkdir -p gpurun_out
for seed in 1 2 3; do
  (cd variants/r3tree && python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --seed $seed > ../../gpurun_out/abs_r3_$seed.json 2>/dev/null)
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --warm-start local --seed $seed > gpurun_out/abs_r4local_$seed.json 2>/dev/null
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --seed $seed > gpurun_out/abs_r4best_$seed.json 2>/dev/null
  for t in r3 r4local r4best; do python3 - <<PY
import json
d=json.loads(open("gpurun_out/abs_${t}_$seed.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("$t seed $seed: %.2f ms/step (E %.2f + M %.2f); launches strip %d comp %d energy %d; swept %.2fG" % (d["ms_per_step"], d["estep_ms"], d["mstep_ms"], k["strip"]["launches"], k["component"]["launches"], k["energy"]["launches"], d["roofline_limiter"]["swept_cells"]/1e9))
PY
  done
done
