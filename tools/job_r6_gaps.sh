#!/bin/bash
# GPU box: the energy gaps of every live-gco parity case (grid and k-NN graphs) on this build -> gpurun_out/r6_live_gco_energy_gaps.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests/test_gpu_estep.py -q -m gpu -k "live_gco or no_grid" -s > gpurun_out/r6_gaps_raw.log 2>&1
grep -E "^n [0-9]+ K|^knn n|passed|failed" gpurun_out/r6_gaps_raw.log > gpurun_out/r6_live_gco_energy_gaps.txt
cat gpurun_out/r6_live_gco_energy_gaps.txt | cut -c1-220
