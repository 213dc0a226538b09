#!/bin/bash
# development: is the package's GPU_MAX_HW_QUEUES default picked up (vs. 4 from the shell)?
for q in default 4 default 4; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 python bench.py --steps 10 --warmup 5 > gpurun_out/q2_${q}.json 2> gpurun_out/q2_${q}.err || exit 1
  python - <<P
import json
d=json.loads(open('gpurun_out/q2_${q}.json').read().strip().splitlines()[-1])
print("queues $q: %.2f ms/step, E-step %.2f, M-step %.2f" % (d['ms_per_step'], d['estep_ms'], d['mstep_ms']))
P
done
