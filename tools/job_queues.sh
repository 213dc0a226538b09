#!/bin/bash
# development: whole-genome bench under different numbers of HIP hardware queues / block threads
for q in ${QUEUES:-4 8 12 16}; do
  for t in ${THREADS:-12 16}; do
    GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 10 --warmup 5 --block-threads $t --no-cpu-baseline > gpurun_out/q_${q}_${t}.json 2> gpurun_out/q_${q}_${t}.err || exit 1
    python - <<P
import json
d=json.loads(open('gpurun_out/q_${q}_${t}.json').read().strip().splitlines()[-1])
print("queues $q threads $t: %.2f ms/step, E-step %.2f, M-step %.2f" % (d['ms_per_step'], d['estep_ms'], d['mstep_ms']))
P
  done
done
