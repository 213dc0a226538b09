#!/bin/bash
# GPU box: one-GPU rehearsal of every rank of a WORLD-GPU run (no driver line).  usage: bash tools/job_emulate_only.sh TAG WORLD [bench flags]
TAG=$1; WORLD=$2; shift; shift
mkdir -p gpurun_out
for r in $(seq 0 $((WORLD-1))); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fit --emulate-world $WORLD --emulate-rank $r "$@" \
      > gpurun_out/${TAG}_emu${WORLD}_r${r}.json 2> gpurun_out/${TAG}_emu${WORLD}_r${r}.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/${TAG}_emu${WORLD}_r${r}.json").read().strip().splitlines()[-1])
e=d["emulated"]
print("rank %d: %.2f M nodes, %d blocks + %d tiles: E-step %.2f ms, M-step (%d states) %.2f ms, step %.2f ms; coarse launches %d" % (e["rank"], e["nodes"]/1e6, e["whole_blocks"], len(e["tiles"]), d["estep_ms"], e["mstep_states"], d["mstep_ms"], d["ms_per_step"], d["kernels"]["coarse"]["launches"]))
PY
done
