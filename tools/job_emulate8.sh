#!/bin/bash
# GPU box: the driver's command on one GPU, then a one-GPU rehearsal of every rank of an 8-GPU run of the same workload
# (bench.py --emulate-world 8 --emulate-rank r): what DESIGN.md section 6's projection table is built from.
# usage: bash tools/job_emulate8.sh [TAG] [WORLD] [extra bench flags...]
TAG=${1:-r4}; WORLD=${2:-8}; shift; shift
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/${TAG}_drv.json 2> gpurun_out/${TAG}_drv.err || exit 1
tail -c 300 gpurun_out/${TAG}_drv.json; echo
for r in $(seq 0 $((WORLD-1))); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --emulate-world $WORLD --emulate-rank $r "$@" \
      > gpurun_out/${TAG}_emu${WORLD}_r${r}.json 2> gpurun_out/${TAG}_emu${WORLD}_r${r}.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_emu${WORLD}_r${r}.json").read().strip().splitlines()[-1])
e=d["emulated"]
print("rank %d: %.2f M nodes, %d blocks + tiles %s: E-step %.2f ms, M-step (%d states) %.2f ms, step %.2f ms" % (e["rank"], e["nodes"]/1e6, e["whole_blocks"], e["tiles"], d["estep_ms"], e["mstep_states"], d["mstep_ms"], d["ms_per_step"]))
PY
done
