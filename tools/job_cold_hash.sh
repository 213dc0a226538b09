#!/bin/bash
# GPU box: cold solves (deterministic mode) of three blocks of cfg3 -- labels hash, energy, rounds, time: before / after a kernel change
mkdir -p gpurun_out
for spec in "0 0" "4 0" "9 4" "7 4"; do
  set -- $spec
  PHMRF_DETERMINISTIC=1 python3 tools/cold_trace.py cfg3 $1 $2 2>/dev/null | grep -v amdgpu.ids
done
