#!/bin/bash
# round 5: what the component pass is worth in a WARM solve (chr1-sized block; tools/trace.py K N tol expansions components)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export PHMRF_TRACE_PERT=${PERT:-0.05}
for comp in 1 0; do
  PHMRF_SOLVE_TRACE=1 python3 tools/trace.py 20 4980 1000 1 $comp > gpurun_out/comp_$comp.out 2> gpurun_out/comp_$comp.err
  echo "== components $comp"; grep -E "warm solve" gpurun_out/comp_$comp.out | cut -c1-300
  sed -n '/---- warm/,$p' gpurun_out/comp_$comp.err | grep -E "round|changed by slot" | cut -c1-260
done
