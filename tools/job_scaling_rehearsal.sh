#!/bin/bash
# GPU box: one-GPU rehearsals for DESIGN.md section 6 -- the ranks of 2- and 4-GPU runs of cfg3, and the chr1 block as 1, 2, 3
# local row tiles (what the lockstep rounds cost without a network in between)
mkdir -p gpurun_out
bash tools/job_emulate_only.sh r4 2 2>&1 | tee gpurun_out/r4_emu2.log
bash tools/job_emulate_only.sh r4 4 2>&1 | tee gpurun_out/r4_emu4.log
for p in 1 2 3; do
  python3 bench.py --workload cfg3-chr1 --steps 20 --warmup 5 --no-cpu-baseline --no-fit --tile-parts $p > gpurun_out/r4_chr1_tiles$p.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4_chr1_tiles$p.json").read().strip().splitlines()[-1])
print("chr1 block as $p tile(s) on one GPU: E-step %.2f ms, M-step %.2f ms; strip launches %d" % (d["estep_ms"], d["mstep_ms"], d["kernels"]["strip"]["launches"]))
PY
done 2>&1 | tee gpurun_out/r4_chr1_tiles.log
